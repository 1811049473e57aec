#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_large_scene.py tests/test_gpu_parity.py tests/test_octree_build.py tests/test_film_filters.py -m gpu -q -x -k "c5 or pm or photon or film or resident" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
WORKLOAD=c5 SQRTSPP=8 STEPS=3 bash tools/ab_builds.sh r04opt6 r04pm r04opt6 r04pm 2>&1 | cut -c1-170 | tee $O/ab_c5.log
