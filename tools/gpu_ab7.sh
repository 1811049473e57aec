#!/bin/bash
O=gpurun_out/r02_ab7
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python tools/ab_probe.py c3 --sqrtspp 8 --steps 2 "w16:" "w12:MCRT_TRACE_WAVES=12" "w8:MCRT_TRACE_WAVES=8" "w4:MCRT_TRACE_WAVES=4" "stack8:MCRT_TRACE_STACK=8" "stack4:MCRT_TRACE_STACK=4" "lds64k:MCRT_TRACE_LDS=65536" ) > $O/c3_waves.log 2>&1
grep -v "amdgpu.ids" $O/c3_waves.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['variant'], d.get('ms_best'), d.get('same_bits_as_first'), d.get('error'))"
