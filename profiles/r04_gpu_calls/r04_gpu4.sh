#!/bin/bash
# round 4, GPU call 4: the tests that failed in call 3 + the new ones, then the one-GPU strong-scaling rehearsal of C3 / C4 / C5
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_deep_tree.py tests/test_nested_media.py tests/test_bvh_build.py tests/test_gpu_large_scene.py -m gpu -q -rA > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log; grep "^FAILED\|^ERROR" $O/pytest_subset.log | head
timeout 600 python tools/shard_probe.py c3 --reps 1 > $O/shard_c3.json 2> $O/shard_c3.err; python -c "
import json,sys
for f in ['c3']:
    r=json.load(open('gpurun_out/r04d/shard_%s.json'%f)); print(f, {n:(v['slowest_shard_ms'],v['predicted_speedup']) for n,v in r['N'].items()})"
timeout 900 python tools/shard_probe.py c4 --sqrtspp 16 --reps 1 > $O/shard_c4.json 2> $O/shard_c4.err; python -c "
import json
r=json.load(open('gpurun_out/r04d/shard_c4.json')); print('c4', {n:(v['slowest_shard_ms'],v['predicted_speedup']) for n,v in r['N'].items()})"
timeout 900 python tools/shard_probe.py c5 --reps 1 > $O/shard_c5.json 2> $O/shard_c5.err; python -c "
import json
r=json.load(open('gpurun_out/r04d/shard_c5.json')); print('c5', {n:(v['slowest_shard_ms'],v['predicted_speedup'],v.get('predicted_speedup_with_photon_pass')) for n,v in r['N'].items()}); print(r.get('photon_pass'))"
tail -3 $O/*.err
