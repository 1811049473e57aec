#!/bin/bash
O=gpurun_out/r04v; mkdir -p $O
WORKLOAD=pm SQRTSPP=2 STEPS=5 bash tools/ab_builds.sh r04pos r04hyb2 r04pos r04hyb2 2>&1 | cut -c1-130 | tee $O/ab_pm.log
WORKLOAD=c5 SQRTSPP=8 STEPS=2 EMISSIONS=1e7 bash tools/ab_builds.sh r04spill2 r04hyb2 r04spill2 r04hyb2 2>&1 | cut -c1-130 | tee $O/ab_c5.log
timeout 600 python -m pytest tests/test_knn_large_k.py tests/test_gpu_parity.py -m gpu -q -k "knn or large_k or pm or photon" 2>&1 | tail -2
