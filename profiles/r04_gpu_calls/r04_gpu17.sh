#!/bin/bash
# emission with the sizing pilot: tests, then the photon pass of C5 (1e8 paths) and of pm as bench.py runs them
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests/test_photon_emission.py tests/test_gpu_large_scene.py tests/test_nested_media.py -m gpu -q -k "emission or photon or emit" 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/photon_pass.log
import importlib, sys, argparse, json
sys.path.insert(0, ".")
import bench
m = importlib.import_module("monte-carlo-ray-tracer_amd"); tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
for name in ("c5", "pm"):
    args = argparse.Namespace(emissions=None, host_octree=False, child_frame=False)
    wl = bench.setup_workload(name, args, m, tiling, 0, 1, 0, None, sqrtspp=1)
    e = wl.emit_info
    print(name, json.dumps({k: e[k] for k in ("photon_pass_s", "photon_pass_first_call_s", "kernel_ms", "octree_build_s", "global_photons", "caustic_photons")}))
    wl.ctx.close()
PY
