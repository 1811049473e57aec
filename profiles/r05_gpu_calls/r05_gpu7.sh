#!/bin/bash
# round 5, call 7: C5 again, round 4's library against the current one - Trav is round 4's struct again (the lean visit's seven words
# are their own LeanRay, held by the trace kernel only), the culled leaf step is gone; then the GPU tier on this library
mkdir -p gpurun_out/r05
date
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 STEPS=2 bash tools/ab_builds.sh r04 new r04 new 2>&1 | tee gpurun_out/r05/ab_c5_builds2.log
WORKLOAD=c3 SQRTSPP=8 STEPS=2 bash tools/ab_builds.sh r04 new 2>&1 | tee -a gpurun_out/r05/ab_c5_builds2.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r05/pytest_call7.log
date
