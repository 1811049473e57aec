#!/bin/bash
O=gpurun_out/r02_ab7
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python tools/ab_probe.py c5 --steps 1 "l32_i8:" "l24:MCRT_PM_LEAF=24" "l16:MCRT_PM_LEAF=16" "l40:MCRT_PM_LEAF=40" "l48_i4:MCRT_PM_LEAF=48,MCRT_PM_MININNER=4" "i16:MCRT_PM_MININNER=16" "i4:MCRT_PM_MININNER=4" "l24_i12:MCRT_PM_LEAF=24,MCRT_PM_MININNER=12") > $O/c5_gates.log 2>&1
grep -v "amdgpu.ids" $O/c5_gates.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['variant'], d.get('ms_best'), d.get('same_bits_as_first'))"
