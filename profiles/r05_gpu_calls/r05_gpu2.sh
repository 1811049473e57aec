#!/bin/bash
# round 5, call 2: the whole GPU tier on the tree with the lean trace visit (MCRT_WF_LEAN, default on), refSinCosF in Photon::dir, the
# RCCL rehearsal; then the lean visit against round 4's on C3 / C4 / spaceship (same process, alternating)
mkdir -p gpurun_out/r05
date
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r05/pytest_call2.log
for spec in "c3 8" "c4 4" "spaceship 8"; do
  set -- $spec
  timeout 400 python tools/ab_probe.py $1 --sqrtspp $2 --steps 2 "r4:MCRT_WF_LEAN=0" "lean:" "lean_multi:MCRT_WF_LEAN=2" "r4:MCRT_WF_LEAN=0" "lean:" "lean_multi:MCRT_WF_LEAN=2" 2>&1 | grep '^{' | cut -c1-220 | sed "s/^/$1 /" | tee -a gpurun_out/r05/ab_trace_lean.log
done
date
