import importlib, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
m = importlib.import_module("monte-carlo-ray-tracer_amd")
img = m.SceneImage(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hexagon_room_pm.mcrt"))
s = img.scene
ctx = m.Context(0); ctx.upload_scene(s)
t=time.time(); em = ctx.emit_photons(1e6, 10.0, 0x12345678); t_emit=time.time()-t
t=time.time()
g = m.PhotonMap(em["global_"][0], s.bb_min[:], s.bb_max[:], 200); c = m.PhotonMap(em["caustic"][0], s.bb_min[:], s.bb_max[:], 200)
t_build=time.time()-t
ctx.upload_photons(g.desc, c.desc, 50, False)
cam = img.camera; cam.width, cam.height, cam.sqrtspp = 1920, 1080, 2
out, st = ctx.sample_image(cam, 0x12345678, m.INTEGRATOR_PHOTON_MAPPER)
out, st = ctx.sample_image(cam, 0x12345678, m.INTEGRATOR_PHOTON_MAPPER)
print("photons g=%d c=%d emit %.2fs (kernel %.1f ms) build %.2fs" % (g.desc.num_photons, c.desc.num_photons, t_emit, em["kernel_ms"], t_build))
print("PM frame 1080p@4spp: kernel %.1f ms, rays %d (%.1f Mray/s), knn %d (%.2f M searches/s)" % (st["kernel_ms"], st["rays"], st["rays"]/st["kernel_ms"]/1e3, st["knn_searches"], st["knn_searches"]/st["kernel_ms"]/1e3))
