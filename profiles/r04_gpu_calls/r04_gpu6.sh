#!/bin/bash
# round 4, GPU call 6: trace kernel with one pop site per iteration, prefix-sum item allocation, root record in LDS: tests + A/B of builds
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scene.py tests/test_deep_tree.py tests/test_gpu_multiprocess.py -m gpu -q -x > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
WORKLOAD=c3 SQRTSPP=8 STEPS=3 bash tools/ab_builds.sh r04base r04opt r04base r04opt 2>&1 | cut -c1-150 | tee $O/ab_c3_64.log
WORKLOAD=c3 SQRTSPP=32 STEPS=1 bash tools/ab_builds.sh r04base r04opt 2>&1 | cut -c1-150 | tee $O/ab_c3_full.log
WORKLOAD=c4 SQRTSPP=8 STEPS=2 bash tools/ab_builds.sh r04base r04opt 2>&1 | cut -c1-150 | tee $O/ab_c4.log
MCRT_COUNT_TESTS=1 timeout 300 python tools/ab_probe.py c3 --steps 1 "count:" > $O/trace_stats_c3_full.log 2>&1; grep "mcrt trace" $O/trace_stats_c3_full.log | head -1
