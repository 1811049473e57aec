#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scene.py tests/test_deep_tree.py -m gpu -q -x > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
WORKLOAD=c3 SQRTSPP=8 STEPS=3 bash tools/ab_builds.sh r04opt6 r04opt7 r04opt6 r04opt7 2>&1 | cut -c1-150 | tee $O/ab_c3_64.log
WORKLOAD=c4 SQRTSPP=8 STEPS=2 bash tools/ab_builds.sh r04opt6 r04opt7 2>&1 | cut -c1-150 | tee $O/ab_c4.log
MCRT_COUNT_TESTS=1 timeout 300 python tools/ab_probe.py c3 --steps 1 "count:" > $O/trace_stats.log 2>&1; grep "mcrt trace" $O/trace_stats.log | sort -u
