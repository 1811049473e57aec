"""The opt-in TOLERANCE build (libmcrt_hip_tol.so: -ffp-contract=fast + the platform's libm, monte-carlo-ray-tracer_amd/build.py) at
BASELINE.json's bar: per-pixel radiance within 1e-4 relative of the reference's golden frames. It is not the reference's bits - FP64
multiply-adds are fused - so the bar here is the contract's, with the outlier budget stated: a path tracer is chaotic (a last-bit change
that flips one russian-roulette or hit / miss decision replaces a whole path), so a frame may hold a few pixels whose 16-spp mean moved by
one path's worth; at most max(2, 0.2 %) of a frame's pixels beyond 1e-4, none non-finite. The exact build (the default, every other GPU
test) keeps bit equality. The library is chosen at import (MCRT_TOLERANCE_BUILD=1), hence the child process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
LIB_TOL = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "libmcrt_hip_tol.so")

CHILD = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import camera_for, golden_path, load_radiance, rel_error
m = importlib.import_module("monte-carlo-ray-tracer_amd")
assert m.TOLERANCE_BUILD and m.LIB_PATH.endswith("libmcrt_hip_tol.so"), m.LIB_PATH
manifest = json.load(open(os.path.join(%(root)r, "tests", "golden", "manifest.json")))
ctx = m.Context(0)
out = {}
for name, integ in %(cases)r:
    case = manifest["cases"][name]
    img = m.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    if integ:
        ctx.upload_photons(img.photons(0), img.photons(1), int(img.param("k_nearest_photons")), bool(img.param("direct_visualization")))
    r = [x for x in case["renders"] if tuple(x["rows"]) == (0, x["height"])][0]
    frame, st = ctx.sample_image(camera_for(img, r), manifest["seed"], integ)
    ref = load_radiance(r)
    rel = rel_error(frame, ref).max(axis=2)
    out[name] = dict(max_rel=float(rel.max()), p999=float(np.quantile(rel, 0.999)), outliers=int((rel > 1e-4).sum()), pixels=int(rel.size),
                     finite=bool(np.isfinite(frame).all()), bit_identical=bool(np.array_equal(frame, ref)), kernel=int(st["kernel_id"]))
ctx.close()
print("RESULT " + json.dumps(out))
"""
CASES = [("hexagon_room", 0), ("hexagon_room_ggx", 0), ("ggx_test", 0), ("metals", 0), ("ior_test", 0), ("veach_mis", 0), ("coffee_maker_qsah", 0),
         ("dragon_room", 0), ("hexagon_room_pm", 1)]


def test_tolerance_build_renders_the_goldens_within_the_contract():
    if not os.path.exists(LIB_TOL):
        pytest.skip("libmcrt_hip_tol.so not built (python __graft_entry__.py build)")
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, cases=CASES)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MCRT_TOLERANCE_BUILD="1"), cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for name, r in res.items():
        print("%s (tolerance build): max rel %.3e, 99.9th pct %.3e, pixels beyond 1e-4: %d / %d, reference's bits: %s"
              % (name, r["max_rel"], r["p999"], r["outliers"], r["pixels"], r["bit_identical"]))
        assert r["finite"], name
        assert r["outliers"] <= max(2, int(0.002 * r["pixels"])), "%s: %d pixels beyond 1e-4" % (name, r["outliers"])
        assert r["p999"] <= 1e-4, name


def test_default_import_is_the_exact_library(pkg):
    assert not pkg.TOLERANCE_BUILD and pkg.LIB_PATH.endswith("libmcrt_hip.so")
