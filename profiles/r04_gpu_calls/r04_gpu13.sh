#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_large_k.py tests/test_gpu_parity.py -m gpu -q -rA -k "knn or pm or photon or large_k" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "max rel|FAILED|Error" $O/pytest.log | head -20
WORKLOAD=pm SQRTSPP=2 STEPS=5 bash tools/ab_builds.sh r04pos r04spill r04pos r04spill 2>&1 | cut -c1-150 | tee $O/ab_pm.log
WORKLOAD=c5 SQRTSPP=8 STEPS=2 EMISSIONS=1e7 bash tools/ab_builds.sh r04pos r04spill 2>&1 | cut -c1-150 | tee $O/ab_c5.log
