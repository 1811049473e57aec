#!/bin/bash
O=gpurun_out/r02_ab7
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python tools/ab_probe.py c3 --sqrtspp 8 --steps 2 "blockpush:" ) > $O/c3_blockpush.log 2>&1
grep -v "amdgpu.ids" $O/c3_blockpush.log | tail -1
python -m pytest tests -m gpu -q -x -k "wavefront or film or full_size or pipeline" 2>&1 | tail -2
