#!/usr/bin/env python3
"""How much of the trace kernel's work is the TREE's fault? CPU-only probe (host build of the device walk, tests/emu).

The GPU walk's cost per ray is (block visits x ~200 wave instructions) + (leaf steps x ~150). Both counts depend on the hierarchy the
walk is given. This probe rebuilds the BVH of a scene image with the builders the library has (mcrt_bvh_build_sah: the reference's
binary / quaternary rules at any bin count) and counts, with the host build of the device walk (emu_trace_counts), box tests and
primitive tests per ray on the same ray set: rays leaving random surface points in cosine-distributed directions (what bounce and
shadow rays look like) plus camera rays.

  python tools/tree_quality_probe.py [image] [n_rays]
"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def surface_rays(sc, n, rng):
    ns = sc.num_surfaces
    kind = np.ctypeslib.as_array(sc.surf_kind, shape=(ns,))
    v = np.ctypeslib.as_array(sc.surf_v, shape=(ns, 9))
    tri = np.nonzero(kind == 0)[0]
    area = np.ctypeslib.as_array(sc.surf_area, shape=(ns,))[tri]
    pick = tri[rng.choice(len(tri), size=n, p=area / area.sum())]
    v0, v1, v2 = v[pick, 0:3], v[pick, 3:6], v[pick, 6:9]
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    p = (1 - r1)[:, None] * v0 + (r1 * (1 - r2))[:, None] * v1 + (r1 * r2)[:, None] * v2
    nrm = np.cross(v1 - v0, v2 - v0)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1), 1e-300)[:, None]
    nrm *= np.where(rng.random(n) < 0.5, -1.0, 1.0)[:, None]
    # cosine-distributed direction around nrm
    a = np.where(np.abs(nrm[:, 0:1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t = np.cross(nrm, a)
    t /= np.linalg.norm(t, axis=1)[:, None]
    b = np.cross(nrm, t)
    u1, u2 = rng.random(n), rng.random(n)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    d = (r * np.cos(phi))[:, None] * t + (r * np.sin(phi))[:, None] * b + np.sqrt(1 - u1)[:, None] * nrm
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.ascontiguousarray(p + 1e-7 * nrm), np.ascontiguousarray(d)


def main():
    image = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "_ref", "images", "metal_bunnies_c3.mcrt")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    import conftest
    emu = conftest.load_emu()
    emu.emu_trace_counts.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    img = m.SceneImage(image)
    sc = img.scene
    rng = np.random.default_rng(7)
    o, d = surface_rays(sc, n, rng)
    print("scene: %d surfaces, %d nodes (as flattened by the reference); %d rays" % (sc.num_surfaces, sc.num_nodes, n))

    def count(desc, label, build_s=0.0):
        out = (C.c_uint64 * 2)()
        for policy in (1,):
            t0 = time.time()
            rc = emu.emu_trace_counts(C.byref(desc), n, o.ctypes.data, d.ctypes.data, policy, out)
            print("%-44s nodes %8d  box tests/ray %6.2f  prim tests/ray %6.2f  (rc %d, build %.1f s, walk %.1f s)"
                  % (label, desc.num_nodes, out[0] / n, out[1] / n, rc, build_s, time.time() - t0), flush=True)

    count(sc, "image's own tree")
    for kind, bins in (("quaternary_sah", 8), ("quaternary_sah", 16), ("quaternary_sah", 32), ("binary_sah", 16), ("binary_sah", 64)):
        t0 = time.time()
        try:
            bvh = m.Bvh(sc, kind=kind, bins_per_axis=bins)
            owned = bvh.apply(sc)
        except Exception as ex:
            print(kind, bins, "failed:", ex)
            continue
        count(owned.desc, "%s bins %d" % (kind, bins), time.time() - t0)


if __name__ == "__main__":
    main()
