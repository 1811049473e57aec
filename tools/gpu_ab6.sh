#!/bin/bash
O=gpurun_out/r02_ab6
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/ab_probe.py pm --steps 3 "warm:" ) > $O/pm_warm.log 2>&1
grep -v "amdgpu.ids" $O/pm_warm.log | tail -2
( timeout 900 python tools/ab_probe.py c5 --steps 1 "warm:" ) > $O/c5_warm.log 2>&1
grep -v "amdgpu.ids" $O/c5_warm.log | tail -2
python -m pytest tests -m gpu -q -k "pm or photon or knn or c5" 2>&1 | tail -3
