#!/bin/bash
O=gpurun_out/r02_ab7
mkdir -p $O
export TMPDIR=/tmp
for w in c3 spaceship; do
( timeout 900 python tools/ab_probe.py $w --sqrtspp 8 --steps 2 "fp32setup:" ) > $O/${w}.log 2>&1
grep -v "amdgpu.ids" $O/${w}.log | tail -1
done
( timeout 900 python tools/ab_probe.py c5 --steps 1 "fp32setup:" ) > $O/c5.log 2>&1
grep -v "amdgpu.ids" $O/c5.log | tail -1
python -m pytest tests -m gpu -q -x -k "full_size or large or traversal or intersect or spaceship or wavefront" 2>&1 | tail -3
