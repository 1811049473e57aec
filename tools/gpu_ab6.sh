#!/bin/bash
O=gpurun_out/r02_ab6
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/ab_probe.py pm --steps 3 "trig:" ) > $O/pm_trig.log 2>&1
grep -v "amdgpu.ids" $O/pm_trig.log | tail -1
( timeout 900 python tools/ab_probe.py c5 --steps 2 "trig:" ) > $O/c5_trig.log 2>&1
grep -v "amdgpu.ids" $O/c5_trig.log | tail -1
python -m pytest tests -m gpu -q -x -k "pm or photon or knn or c5" 2>&1 | tail -2
