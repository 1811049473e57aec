#!/bin/bash
# C2: samples per work unit against units per lane (whole frame, and what one rank of 8 renders)
O=gpurun_out/r04r; mkdir -p $O
timeout 300 python tools/ab_probe.py c2 --steps 3 "auto:" "u4:MCRT_CHUNKS=4" "u8:MCRT_CHUNKS=8" "u16:MCRT_CHUNKS=16" "u32:MCRT_CHUNKS=32" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c2 full', d['variant'], d['ms_best'], d['Mray_s'], d['same_bits_as_first'])" | tee $O/chunks.log
for u in auto 2 4 8 16 32 64; do
  if [ $u = auto ]; then unset MCRT_CHUNKS; else export MCRT_CHUNKS=$u; fi
  timeout 300 python tools/shard_probe.py c2 --sqrtspp 16 --n 2,4,8 --shards ends --reps 3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2 shards, units per pixel $u:', {n: v['slowest_shard_ms'] for n, v in r['N'].items()})" | tee -a $O/chunks.log
done
