#!/bin/bash
O=gpurun_out/r02_ab4
mkdir -p $O
export TMPDIR=/tmp
( MCRT_COUNT_TESTS=1 timeout 900 python tools/ab_probe.py c5 --steps 1 "count:" ) > $O/c5_count.log 2>&1
grep -v "^\[mcrt phase\]" $O/c5_count.log | tail -6
( MCRT_COUNT_TESTS=1 timeout 900 python tools/ab_probe.py pm --steps 1 "count:" ) > $O/pm_count.log 2>&1
grep -v "^\[mcrt phase\]" $O/pm_count.log | tail -6
