#!/bin/bash
O=gpurun_out/r02_ab5
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python tools/ab_probe.py c3 --sqrtspp 8 --steps 2 "hoist:" ) > $O/c3_hoist.log 2>&1
grep -v "amdgpu.ids" $O/c3_hoist.log | tail -3
( timeout 900 python tools/ab_probe.py c4 --sqrtspp 8 --steps 2 "hoist:" ) > $O/c4_hoist.log 2>&1
grep -v "amdgpu.ids" $O/c4_hoist.log | tail -3
