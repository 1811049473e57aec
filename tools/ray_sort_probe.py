#!/usr/bin/env python3
"""How much would the trace kernel gain from rays sorted for coherence? Builds secondary rays of a workload's scene
(camera rays -> hit -> cosine-distributed bounce off the geometric normal, twice) and times mcrt_intersect's trace kernel
(MCRT_OP_TIME=1 prints it) on the SAME ray set in different orders: as generated (pixel order), shuffled (what a slot pool
looks like after a few bounces), and sorted by a few candidate keys. The hits must not depend on the order.
    python tools/ray_sort_probe.py [workload]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
os.environ["MCRT_OP_TIME"] = "1"


def morton3(q, bits):
    """q: (n, 3) integer cells < 2^bits -> interleaved code."""
    code = np.zeros(q.shape[0], dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code


def bounce(rng, o, d, t, surf, v0, e):
    hit = np.isfinite(t) & (t > 0) & (surf != 0xFFFFFFFF)
    o, d, t, surf = o[hit], d[hit], t[hit], surf[hit]
    p = o + t[:, None] * d
    n = np.cross(e[surf, :3], e[surf, 3:])
    ln = np.linalg.norm(n, axis=1)
    ok = ln > 0
    p, d, n = p[ok], d[ok], n[ok] / ln[ok, None]
    n[np.einsum("ij,ij->i", n, d) > 0] *= -1.0
    # cosine-distributed direction around n
    u1, u2 = rng.random(p.shape[0]), rng.random(p.shape[0])
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    tx = np.cross(n, a)
    tx /= np.linalg.norm(tx, axis=1)[:, None]
    ty = np.cross(n, tx)
    nd = tx * (r * np.cos(phi))[:, None] + ty * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0.0, 1 - u1))[:, None]
    nd /= np.linalg.norm(nd, axis=1)[:, None]
    return p + 1e-7 * n, nd


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    import bench

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")

    class A:
        emissions = 1e6
        host_octree = False

    wl = bench.setup_workload(name, A, m, tiling, 0, 1, 0, None, sqrtspp=1)
    sc = wl.img.scene
    ns = sc.num_surfaces
    e = np.ctypeslib.as_array(sc.surf_e, (ns, 6)).copy()
    v0 = None
    cam = wl.full
    W, H = 1920, 1080
    eye = np.array(cam.eye[:])
    fw, lf, up = np.array(cam.forward[:]), np.array(cam.left[:]), np.array(cam.up[:])
    px = cam.sensor_width / W
    ys, xs = np.mgrid[0:H, 0:W]
    d = fw[None, :] * cam.focal_length + lf[None, :] * ((W / 2 - xs.ravel() - 0.5) * px)[:, None] + up[None, :] * ((H / 2 - ys.ravel() - 0.5) * px)[:, None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    o = np.repeat(eye[None, :], d.shape[0], axis=0)
    rng = np.random.default_rng(7)
    sets = {}
    for b in range(3):
        print("bounce %d: %d rays" % (b, o.shape[0]), file=sys.stderr, flush=True)
        t, surf, _ = wl.ctx.intersect(o, d)
        if b > 0:
            sets[b] = (o, d, t)
        o, d = bounce(rng, o, d, t, surf, v0, e)
    for b, (o, d, t_ref) in sets.items():
        n = o.shape[0]
        lo, hi = np.array(sc.bb_min[:]), np.array(sc.bb_max[:])
        cell10 = np.clip(((o - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
        octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
        dcell = np.clip(((d + 1) * 0.5 * 8).astype(np.int64), 0, 7)  # 3 bits per axis of direction
        m30 = morton3(cell10, 10)
        m15 = morton3(cell10 >> 5, 5)
        m18 = morton3(cell10 >> 4, 6)
        md9 = morton3(dcell, 3)
        orders = {
            "as generated": np.arange(n),
            "shuffled": rng.permutation(n),
            "origin morton30": np.argsort(m30, kind="stable"),
            "octant | origin morton30": np.argsort((octant << np.uint64(30)) | m30, kind="stable"),
            "origin morton15 | octant | origin low": np.argsort((m15 << np.uint64(18 + 3)) | (octant << np.uint64(18)) | (m30 & np.uint64((1 << 15) - 1)), kind="stable"),
            "origin morton18 | direction morton9": np.argsort((m18 << np.uint64(9)) | md9, kind="stable"),
            "direction morton9 | origin morton30": np.argsort((md9 << np.uint64(30)) | m30, kind="stable"),
        }
        for label, perm in orders.items():
            print("== bounce %d, %s" % (b, label), file=sys.stderr, flush=True)
            for rep in range(2):
                t, surf, _ = wl.ctx.intersect(o[perm], d[perm])
            assert np.array_equal(t, t_ref[perm]), "hits depend on the order?"


if __name__ == "__main__":
    main()
