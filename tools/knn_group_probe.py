#!/usr/bin/env python3
"""knnGroupKernel (four queries per wave, MCRT_KNN_GROUPS=1) against knnWaveKernel: same results, kernel time of both
(MCRT_KNN_TIME=1) on the hexagon_room photon maps, coherent and random queries."""
import importlib, os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
os.environ["MCRT_KNN_TIME"] = "1"
m = importlib.import_module("monte-carlo-ray-tracer_amd")
img = m.SceneImage(os.path.join(ROOT, "tests", "golden", "hexagon_room_pm.mcrt"))
s = img.scene
ctx = m.Context(0); ctx.upload_scene(s)
em = ctx.emit_photons(float(sys.argv[1]) if len(sys.argv) > 1 else 1e6, 10.0, 0x12345678)
maps = [m.PhotonMap(em[key][0], s.bb_min[:], s.bb_max[:], 200, ctx=ctx) for key in ("global_", "caustic")]
ctx.upload_photons(maps[0].desc, maps[1].desc, 50, False)
rng = np.random.default_rng(1)
n = 1 << 20
for which in (0, 1):
    d = maps[which].desc
    ph = np.ctypeslib.as_array(d.photons, (d.num_photons, 8))
    start = rng.integers(0, d.num_photons - n // 4)
    coherent = np.repeat(ph[start:start + n // 4, 3:6].astype(np.float64), 4, axis=0) + rng.normal(scale=2e-3, size=(n, 3))
    scattered = ph[rng.integers(0, d.num_photons, n), 3:6].astype(np.float64) + rng.normal(scale=2e-3, size=(n, 3))
    for name, pts in (("coherent", coherent), ("random", scattered)):
        for k in (50, 7, 64):
            res = {}
            for mode in ("0", "1"):
                os.environ["MCRT_KNN_GROUPS"] = mode
                for rep in range(2):
                    print("map %d (%d photons) %s queries, k = %d, groups %s:" % (which, d.num_photons, name, k, mode), flush=True)
                    res[mode] = ctx.knn(which, np.ascontiguousarray(pts), k)
            for a, b, what in zip(res["0"], res["1"], ("count", "index", "d2")):
                bad = np.nonzero(np.any(np.atleast_2d(a.reshape(n, -1) != b.reshape(n, -1)), axis=1))[0]
                print("   %s: %d of %d queries differ%s" % (what, bad.size, n, (" (first: %d)" % bad[0]) if bad.size else ""), flush=True)
