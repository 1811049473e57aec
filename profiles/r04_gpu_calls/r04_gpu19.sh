#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O
WORKLOAD=pm SQRTSPP=2 STEPS=5 bash tools/ab_builds.sh r04pos r04spill r04spill2 r04cur r04pos r04spill2 r04cur 2>&1 | cut -c1-130 | tee $O/ab_pm.log
WORKLOAD=c5 SQRTSPP=8 STEPS=2 EMISSIONS=1e7 bash tools/ab_builds.sh r04spill2 r04cur 2>&1 | cut -c1-130 | tee $O/ab_c5.log
WORKLOAD=spaceship SQRTSPP=2 STEPS=3 bash tools/ab_builds.sh r04spill2 r04cur 2>&1 | cut -c1-130 | tee $O/ab_spaceship_small.log
timeout 300 python tools/shard_probe.py c2 --sqrtspp 16 --n 1,2,4,8 --shards ends --reps 3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2 shards, new unit rule:', {n: (v['slowest_shard_ms'], v['predicted_speedup']) for n, v in r['N'].items()})" | tee $O/c2_shards.log
