#!/bin/bash
# photon positions as their own array (12 B per scanned photon instead of 32): tests on the new build, then A/B against the round's final build
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large_scene.py tests/test_gpu_parity.py tests/test_octree_build.py tests/test_photon_emission.py -m gpu -q -x -k "c5 or pm or photon or knn or octree" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
WORKLOAD=pm SQRTSPP=2 STEPS=5 bash tools/ab_builds.sh r04final r04pos r04final r04pos 2>&1 | cut -c1-170 | tee $O/ab_pm.log
WORKLOAD=c5 SQRTSPP=8 STEPS=3 EMISSIONS=1e7 bash tools/ab_builds.sh r04final r04pos r04final r04pos 2>&1 | cut -c1-170 | tee $O/ab_c5.log
timeout 300 python tools/ab_probe.py spaceship --steps 3 "sm:" "wf:MCRT_KERNEL=wf" 2>&1 | cut -c1-200 | tee $O/spaceship_wf.log
