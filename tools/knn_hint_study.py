#!/usr/bin/env python3
"""CPU study for the next round (no GPU needed): how much work would a photon search save if it started from the bound
that a NEIGHBOURING search already proved?

For queries q, q' of the same map, r_k(q') <= r_k(q) + |q - q'| (the k photons within r_k(q) of q are within that distance
of q'), so a search may start with max_distance2 = (r_k(q) + |q - q'|)^2 instead of infinity and still return exactly the
reference's k photons. The wave-cooperative search of renderKernelPM serves the 64 lanes' queries one after the other, so
the previous results of the same wave are at hand for free.

Method: the oracle emits a hexagon_room photon map of bench.py's `pm` size on the CPU, builds the octrees with the product's
host builder, renders a crop single-threaded while recording every search (map, position, pixel, sample), replays the
searches in the order a wave would meet them (pixels in 8x8 tiles, a pixel's samples in order) with hints from the last H
searches of the same map, and compares octants visited / photons scanned / results. Prints one JSON line."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    emissions = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
    width, height, sqrtspp = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (192, 108, 2)))
    rows = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else None   # full-width rows of a larger frame
    pkg = importlib.import_module("monte-carlo-ray-tracer_amd")
    import oracle_lib
    L = oracle_lib.lib()
    vp = C.c_void_p
    L.oracle_knn_recorder.argtypes = [vp, C.c_uint64]
    L.oracle_knn_recorded.restype = C.c_uint64
    L.oracle_knn_hinted.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp, vp, vp]
    img = pkg.SceneImage(os.path.join(ROOT, "tests", "golden", "hexagon_room.mcrt"))
    seed, k = 0x12345678, 50
    em = oracle_lib.emit_photons(img, emissions, 10.0, seed)
    sc = img.scene
    maps = [pkg.PhotonMap(em["global_"][0], sc.bb_min[:], sc.bb_max[:], 200), pkg.PhotonMap(em["caustic"][0], sc.bb_min[:], sc.bb_max[:], 200)]

    class WithMaps:  # the SceneImage interface oracle_lib.render uses
        scene = img.scene

        @staticmethod
        def photons(which):
            return maps[which].desc

        @staticmethod
        def param(key):
            return {"k_nearest_photons": k, "direct_visualization": 0}.get(key, 0)

    cam = img.camera.copy()
    cam.width, cam.height, cam.sqrtspp = width, height, sqrtspp
    cap = width * (rows[1] - rows[0] if rows else height) * sqrtspp ** 2 * 8
    rec = np.zeros((cap, 6))
    L.oracle_knn_recorder(rec.ctypes.data, cap)
    oracle_lib.render(WithMaps, cam, seed, pkg.INTEGRATOR_PHOTON_MAPPER, rows=rows, threads=1)
    n = int(L.oracle_knn_recorded())
    L.oracle_knn_recorder(None, 0)
    rec = rec[:n]
    # the order a wave meets the searches: pixels in 8x8 tiles, then sample, then the order inside the path
    pix = rec[:, 4].astype(np.int64)
    x, y = pix % width, pix // width
    tile = (y // 8) * ((width + 7) // 8) + (x // 8)
    order = np.lexsort((np.arange(n), rec[:, 5], (y % 8) * 8 + (x % 8), tile))
    rec = rec[order]
    out = dict(emissions=emissions, photons=[int(m.desc.num_photons) for m in maps], searches=n, k=k, frame="%dx%d@%d" % (width, height, sqrtspp ** 2))
    for which, tag in ((0, "global"), (1, "caustic")):
        q = np.ascontiguousarray(rec[rec[:, 0] == which][:, 1:4])
        m = len(q)
        if m == 0:
            continue

        def run(bound2):
            kth, cnt = np.empty(m), np.empty(m, dtype=np.uint32)
            octs, phs = C.c_uint64(), C.c_uint64()
            L.oracle_knn_hinted(C.byref(maps[which].desc), m, q.ctypes.data, k, bound2.ctypes.data if bound2 is not None else None,
                                kth.ctypes.data, cnt.ctypes.data, C.byref(octs), C.byref(phs))
            return kth, cnt, octs.value, phs.value

        kth0, cnt0, o0, p0 = run(None)
        res = dict(searches=m, octants_per_search=o0 / m, photons_per_search=p0 / m)
        for H in (1, 4, 16, 63):
            # hints from the previous H searches of this map (what the lanes of a wave served before this one know)
            r = np.sqrt(kth0)
            bound = np.full(m, np.inf)
            for h in range(1, H + 1):
                d = np.linalg.norm(q[h:] - q[:-h], axis=1)
                cand = (r[:-h] + d) * (1 + 1e-12) + 1e-300
                full = cnt0[:-h] == k                       # a search that found fewer than k photons proves nothing
                bound[h:] = np.where(full, np.minimum(bound[h:], cand), bound[h:])
            b2 = np.where(np.isfinite(bound), bound * bound, np.finfo(np.float64).max)
            kth, cnt, o, p = run(np.ascontiguousarray(b2))
            assert np.array_equal(cnt, cnt0) and np.array_equal(kth, kth0), "a hinted search changed its result"
            res["hints_%d" % H] = dict(octants_per_search=o / m, photons_per_search=p / m, octants_ratio=o / o0, photons_ratio=p / p0,
                                       median_bound_over_rk=float(np.median(bound[np.isfinite(bound)] / r[np.isfinite(bound)])))
        out[tag] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
