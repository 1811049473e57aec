#!/bin/bash
# round 5, call 6: what slowed C5 down (3.03 -> 3.27 s per frame between round 4's library and call 4's)? Library builds swapped on one
# box (tools/ab_builds.sh), C5 probe at 64 spp, 1e7 x 10 emission paths: round 4's library, the current one, the current one with
# MCRT_LOCKSTEP() as nothing again, the current one with ocml's sinf / cosf in Photon::dir
mkdir -p gpurun_out/r05
date
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 STEPS=2 bash tools/ab_builds.sh r04 new LOCKSTEP_OLD OCML_SINCOSF r04 new LOCKSTEP_OLD OCML_SINCOSF 2>&1 | tee gpurun_out/r05/ab_c5_bisect.log
WORKLOAD=pm SQRTSPP=2 STEPS=3 bash tools/ab_builds.sh r04 new LOCKSTEP_OLD OCML_SINCOSF 2>&1 | tee -a gpurun_out/r05/ab_c5_bisect.log
date
