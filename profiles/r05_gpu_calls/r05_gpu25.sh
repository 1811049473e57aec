mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_large_scene.py -m gpu -q -rA -k "c5 and legacy" > gpurun_out/r05/pytest_c5_legacy.log 2>&1
tail -5 gpurun_out/r05/pytest_c5_legacy.log
