#!/bin/bash
# round 5, call 24: the per-lane photon-mapper kernel with the reference's heap discipline (push_unordered / make_heap / pop_push) and the
# restated sincosf - are its photon-mapped frames the reference's bits? hexagon_room_pm, C5 rows; and every photon / k-NN test again
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_knn_large_k.py tests/test_gpu_large_scene.py tests/test_photon_emission.py tests/test_octree_build.py -m gpu -q -rA -k "photon or knn or c5 or pm" 2>&1 | tail -60 | tee gpurun_out/r05/pytest_gpu_exact_pm.log
