#!/bin/bash
# round 5, call 5: (a) C5's frame regressed in call 4 (3.03 -> 3.27 s): round 4's library against the current one (Trav's lean fields now
# opt-in) on the C5 probe, library builds swapped on one box (tools/ab_builds.sh); (b) the culled shared leaf step (MCRT_WF_CULL=1)
# against the lean form on C3 / C4 / spaceship; (c) the emulation-gated tests of the new form on the device
mkdir -p gpurun_out/r05
date
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 STEPS=2 bash tools/ab_builds.sh r04 new r04 new 2>&1 | tee gpurun_out/r05/ab_c5_builds.log
for spec in "c3 8" "c4 4" "spaceship 8"; do
  set -- $spec
  timeout 400 python tools/ab_probe.py $1 --sqrtspp $2 --steps 2 "lean:MCRT_WF_CULL=0" "cull:MCRT_WF_CULL=1" "lean:MCRT_WF_CULL=0" "cull:MCRT_WF_CULL=1" 2>&1 | grep '^{' | cut -c1-200 | sed "s/^/$1 /" | tee -a gpurun_out/r05/ab_trace_cull.log
done
MCRT_WF_CULL=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_scene.py tests/test_deep_tree.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r05/pytest_cull.log
date
