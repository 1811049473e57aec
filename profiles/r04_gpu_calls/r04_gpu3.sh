#!/bin/bash
# round 4, GPU call 3: GPU tier after the exact shadow queries / per-scene stacks; A/B of library builds (before / after the light
# pre-test) on every kernel family
O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rA > $O/pytest_rA.log 2>&1; tail -3 $O/pytest_rA.log; grep "^FAILED\|^ERROR" $O/pytest_rA.log | head -20
for wl in "c3 8" "spaceship 8" "c2 16" "pm 2" "c5 8"; do set -- $wl
  echo "== $1"; WORKLOAD=$1 SQRTSPP=$2 STEPS=3 bash tools/ab_builds.sh prepretest new prepretest new 2>&1 | cut -c1-150
done > $O/ab_builds.log 2>&1; cat $O/ab_builds.log
