#!/bin/bash
# round 5, call 20: the trace kernel's gates again, now that a visit is cheaper (lean visit): pending lanes that start a shared leaf
# step (MCRT_WF_LEAF), inner lanes below which it starts anyway (MCRT_WF_MININNER), idle lanes that trigger a refill (MCRT_WF_REFILL),
# waves per workgroup (MCRT_TRACE_WAVES) - C3 probe (64 spp) and C4 probe (16 spp), one process each
mkdir -p gpurun_out/r05
timeout 600 python tools/ab_probe.py c3 --sqrtspp 8 --steps 2 "base:" "leaf12:MCRT_WF_LEAF=12" "leaf20:MCRT_WF_LEAF=20" "leaf24:MCRT_WF_LEAF=24" "mininner4:MCRT_WF_MININNER=4" "mininner16:MCRT_WF_MININNER=16" "refill8:MCRT_WF_REFILL=8" "refill24:MCRT_WF_REFILL=24" "waves12:MCRT_TRACE_WAVES=12" "deal7:MCRT_WF_DEAL=7" "base:" 2>&1 | grep '^{' | cut -c1-140 | sed "s/^/c3 /" | tee gpurun_out/r05/ab_trace_gates_lean.log
timeout 600 python tools/ab_probe.py c4 --sqrtspp 4 --steps 2 "base:" "leaf12:MCRT_WF_LEAF=12" "leaf20:MCRT_WF_LEAF=20" "mininner4:MCRT_WF_MININNER=4" "mininner16:MCRT_WF_MININNER=16" "refill8:MCRT_WF_REFILL=8" "refill24:MCRT_WF_REFILL=24" "base:" 2>&1 | grep '^{' | cut -c1-140 | sed "s/^/c4 /" | tee -a gpurun_out/r05/ab_trace_gates_lean.log
# the strong-scaling rehearsal of the headline with this round's kernel form (768-lane renderKernelFlatK): first and last shard of N = 1, 2, 4, 8
timeout 300 python tools/shard_probe.py c2 --shards ends --n 1,2,4,8 > gpurun_out/r05/shard_probe_c2.json 2> gpurun_out/r05/shard_probe_c2.err
tail -c 600 gpurun_out/r05/shard_probe_c2.json
