#!/bin/bash
# round 5, call 29: the strong-scaling rehearsal (what one rank of N renders, first and last shard, on one GPU) with this round's library: C3, C5
mkdir -p gpurun_out/r05
for wl in c3 c5; do
  timeout 500 python tools/shard_probe.py $wl --shards ends --n 1,2,4,8 --reps 1 > gpurun_out/r05/shard_probe_$wl.json 2> gpurun_out/r05/shard_probe_$wl.err
  tail -c 400 gpurun_out/r05/shard_probe_$wl.json; echo
done
