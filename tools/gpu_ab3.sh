#!/bin/bash
O=gpurun_out/r02_ab3
mkdir -p $O
export TMPDIR=/tmp
( MCRT_PROFILE_PHASES=1 timeout 600 python tools/ab_probe.py c2 --steps 1 --sqrtspp 8 "prof:" ) > $O/phases_c2.log 2>&1
tail -12 $O/phases_c2.log
( MCRT_PROFILE_PHASES=1 MCRT_FLAT_CULL=0 timeout 600 python tools/ab_probe.py c2 --steps 1 --sqrtspp 8 "prof_nocull:" ) > $O/phases_c2_nocull.log 2>&1
tail -8 $O/phases_c2_nocull.log
