#!/bin/bash
O=gpurun_out/r02_c5scan
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python tools/ab_probe.py c5 --steps 1 "s16:" "s12:MCRT_PM_STACK=12" "s8:MCRT_PM_STACK=8" "s4:MCRT_PM_STACK=4" "s16b:" ) > $O/c5_stack.log 2>&1
grep -v "amdgpu.ids" $O/c5_stack.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['variant'], d.get('ms_best'), d.get('same_bits_as_first'), d.get('error'))"
