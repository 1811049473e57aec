import importlib, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
m = importlib.import_module("monte-carlo-ray-tracer_amd")
img = m.SceneImage(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hexagon_room_pm.mcrt"))
s = img.scene
ctx = m.Context(0); ctx.upload_scene(s)
em = ctx.emit_photons(float(sys.argv[1]) if len(sys.argv) > 1 else 1e6, 10.0, 0x12345678)
g = m.PhotonMap(em["global_"][0], s.bb_min[:], s.bb_max[:], 200); c = m.PhotonMap(em["caustic"][0], s.bb_min[:], s.bb_max[:], 200)
ctx.upload_photons(g.desc, c.desc, 50, False)
rng = np.random.default_rng(1)
n = 400000
ph = em["global_"][0]
pts = ph[rng.integers(0, len(ph), n), 3:6].astype(np.float64) + rng.normal(scale=0.01, size=(n, 3))
for which in (0, 1):
    t = time.time(); cnt, idx, d2 = ctx.knn(which, pts, 50); dt = time.time() - t
    print("map %d: %d photons, %d octants: %d queries in %.3f s incl. copies -> %.2f M/s" % (which, (g if which == 0 else c).desc.num_photons, (g if which == 0 else c).desc.num_octants, n, dt, n / dt / 1e6))
