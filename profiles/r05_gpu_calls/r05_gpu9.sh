#!/bin/bash
# round 5, call 9: C5 probe, more builds of renderKernelPM (tools/ab_builds.sh): v1 = ocml sinf / cosf + round 4's overflow line, fences
# kept; v2 = refSinCosF as a real (noinline) device function; v3 = v2 + round 4's overflow line; v4 = refSinCosF called in a loop
# over the two angles. Also MCRT_PM_BLOCK=512 on the current build.
mkdir -p gpurun_out/r05
date
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 STEPS=2 bash tools/ab_builds.sh r04 new v1 v2 v3 v4 2>&1 | tee gpurun_out/r05/ab_c5_bisect3.log
timeout 300 python tools/ab_probe.py c5 --sqrtspp 8 --emissions 1e7 --steps 2 "new1024:" "new512:MCRT_PM_BLOCK=512" 2>&1 | grep '^{' | cut -c1-160 | tee -a gpurun_out/r05/ab_c5_bisect3.log
date
