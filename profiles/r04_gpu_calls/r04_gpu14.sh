#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
WORKLOAD=c5 SQRTSPP=8 STEPS=2 EMISSIONS=1e7 bash tools/ab_builds.sh r04pos r04spill2 r04pos r04spill2 2>&1 | cut -c1-150 | tee $O/ab_c5.log
timeout 600 python -m pytest tests/test_knn_large_k.py tests/test_gpu_parity.py -m gpu -q -k "knn or large_k" 2>&1 | tail -2
