mkdir -p gpurun_out/r05
timeout 500 python tools/shard_probe.py c3 --shards ends --n 1,2,4,8 --reps 2 > gpurun_out/r05/shard_probe_c3.json 2> gpurun_out/r05/shard_probe_c3.err
tail -c 300 gpurun_out/r05/shard_probe_c3.json; echo
