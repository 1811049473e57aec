#!/bin/bash
# round 5, call 8: C5's slowdown, one change at a time on the current tree (tools/ab_builds.sh, C5 probe): ocml sinf / cosf in
# Photon::dir instead of refSinCosF; MCRT_LOCKSTEP() as nothing; the overflow words summed as in round 4; all three (= round 4's
# renderKernelPM source)
mkdir -p gpurun_out/r05
date
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 STEPS=2 bash tools/ab_builds.sh r04 new ocml lockold statsold all3 r04 new ocml lockold statsold all3 2>&1 | tee gpurun_out/r05/ab_c5_bisect2.log
date
