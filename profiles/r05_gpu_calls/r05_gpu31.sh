#!/bin/bash
# round 5, call 31: the strong-scaling rehearsal of C4 (4K @ 1024 spp) on the final library, first and last shard of N = 1, 2, 4, 8, one repetition
# (a warm-up frame at 16 spp first: shard_probe's first timed call would otherwise carry the one-time allocations)
mkdir -p gpurun_out/r05
timeout 200 python tools/ab_probe.py c4 --sqrtspp 4 --steps 1 "base:" > /dev/null 2>&1
timeout 700 python tools/shard_probe.py c4 --shards ends --n 1,2,4,8 --reps 1 > gpurun_out/r05/shard_probe_c4.json 2> gpurun_out/r05/shard_probe_c4.err
tail -c 300 gpurun_out/r05/shard_probe_c4.json; echo
