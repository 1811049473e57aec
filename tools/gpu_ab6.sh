#!/bin/bash
O=gpurun_out/r02_ab6
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "knn" 2>&1 | tail -3
( timeout 600 python tools/ab_probe.py pm --steps 3 "wave:" ) > $O/pm_after_revert.log 2>&1
grep -v "amdgpu.ids" $O/pm_after_revert.log | tail -2
