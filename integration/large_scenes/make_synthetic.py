#!/usr/bin/env python3
"""Deterministic stand-ins for meshes that the reference tree lists in .MISSING_LARGE_BLOBS
(SURVEY.md §8(d)). Pure integer-hash noise, no RNG state, so every machine writes the same bytes.

  bunny.obj   icosphere level 6 (40 962 vertices, 81 920 triangles), centre (0, 0.75, 0), radius 0.7,
              radial value-noise displacement +-0.15; no `vn` lines (metal_bunnies.json sets "smooth",
              so Scene::generateVertexNormals runs, scene/scene.cpp:61-65,325-355)
"""
import os
import sys

import numpy as np


def _hash3(ix, iy, iz, seed):
    h = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) ^ \
        (iz.astype(np.uint64) * np.uint64(83492791)) ^ np.uint64(seed * 2654435761 % (1 << 32))
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0xd168aaad)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0xaf723597)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return (h & np.uint64(0xFFFFFF)).astype(np.float64) / float(1 << 24)  # [0,1)


def value_noise(p, freq, seed):
    q = p * freq + 100.0
    i = np.floor(q).astype(np.int64)
    f = q - i
    f = f * f * (3.0 - 2.0 * f)
    out = np.zeros(len(p))
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out += w * _hash3(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed)
    return out


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    def unit(x):  # explicit IEEE operations only (no BLAS: its summation order is CPU dependent)
        n = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) ** 0.5
        return (x[0] / n, x[1] / n, x[2] / n)

    v = [unit(tuple(float(c) for c in x)) for x in v]
    for _ in range(level):
        cache = {}
        nf = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                v.append(unit((v[a][0] + v[b][0], v[a][1] + v[b][1], v[a][2] + v[b][2])))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int64)


def write_bunny(path, level=6, seed=1):
    v, f = icosphere(level)
    n = 0.55 * value_noise(v, 2.0, seed) + 0.3 * value_noise(v, 5.0, seed + 1) + 0.15 * value_noise(v, 11.0, seed + 2)
    r = 0.7 + 0.30 * (n - 0.5)  # +-0.15
    p = v * r[:, None] + np.array([0.0, 0.75, 0.0])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as out:
        out.write("# synthetic stand-in for data/bunny.obj (tests/large/make_synthetic.py)\n")  # (the file lived there when its md5 was pinned: the line is part of the fingerprint)
        for x, y, z in p:
            out.write("v %.17g %.17g %.17g\n" % (x, y, z))
        for a, b, c in f + 1:
            out.write("f %d %d %d\n" % (a, b, c))
    return len(p), len(f)


if __name__ == "__main__":
    print(write_bunny(sys.argv[1] if len(sys.argv) > 1 else "/tmp/bunny.obj"))
