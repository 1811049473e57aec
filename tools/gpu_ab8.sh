#!/bin/bash
O=gpurun_out/r02_ab8
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python tools/ab_probe.py c2 --steps 2 "sincospi:" ) > $O/c2.log 2>&1
grep -v "amdgpu.ids" $O/c2.log | tail -1
python -m pytest tests/test_gpu_parity.py tests/test_film_filters.py -m gpu -q -x 2>&1 | tail -3
