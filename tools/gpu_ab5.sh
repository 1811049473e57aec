#!/bin/bash
O=gpurun_out/r02_ab5
mkdir -p $O
export TMPDIR=/tmp
( MCRT_WF_DEAL=6 timeout 900 python tools/ray_sort_probe.py c3 ) > $O/sort_c3_deal6.log 2>&1
grep -v "amdgpu.ids" $O/sort_c3_deal6.log | grep -A2 "^==" | grep -v "^--" | awk '/^==/{l=$0; getline; getline; print l " :: " $0}'
