#!/usr/bin/env python3
"""Short timing probe of one workload at reduced spp (GPU box): python tools/sm_probe.py c3 4 [repeat]
Prints Mray/s, rays/path and (with MCRT_COUNT_TESTS=1) the kernel's own box/primitive test counts per ray.
Honours every MCRT_* env knob, so it is the unit of the tuning scripts."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    sqrtspp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    if name in ("c3", "c4", "c5"):
        import make_large
        path = make_large.ensure_image(name)
    elif name == "spaceship":
        path = os.path.join(ROOT, "oracle", "_ref", "images", "spaceship.mcrt")
    else:
        path = os.path.join(ROOT, "tests", "golden", name + ".mcrt")
    img = m.SceneImage(path)
    cam = img.camera
    cam.width, cam.height, cam.sqrtspp = 1920, 1080, sqrtspp
    ctx = m.Context(0)
    ctx.upload_image(img)
    for i in range(repeat):
        out, st = ctx.sample_image(cam, 0x12345678, m.INTEGRATOR_PATH_TRACER)
        line = "%s @%d spp: %.1f Mray/s  kernel %.1f ms  rays/path %.2f  mean %.9f" % (
            name, sqrtspp ** 2, st["rays"] / st["kernel_ms"] / 1e3, st["kernel_ms"], st["rays"] / st["paths"], out.mean())
        if st["node_tests"]:
            line += "  box tests/ray %.2f  prim tests/ray %.2f" % (st["node_tests"] / st["rays"], st["prim_tests"] / st["rays"])
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
