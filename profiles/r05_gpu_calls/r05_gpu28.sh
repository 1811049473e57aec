timeout 600 python -m pytest tests/test_libm.py -m gpu -q 2>&1 | tail -4
