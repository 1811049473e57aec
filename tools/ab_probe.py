#!/usr/bin/env python3
"""A/B runs of one workload of bench.py under different run-time knobs, in ONE process on one GPU (the knobs are read
with getenv at every launch). Usage:
    python tools/ab_probe.py <workload> [--sqrtspp S] [--steps K] [--emissions E] VARIANT [VARIANT ...]
with VARIANT = name:KEY=VAL,KEY=VAL (an empty assignment list = the defaults). Prints one JSON line per variant:
ms per frame (best and mean of K), Mray/s, searches/s, kernel id, frame checksum (must agree across variants)."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--sqrtspp", type=int, default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--emissions", type=float, default=1e6)
    ap.add_argument("--host-octree", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    wl = bench.setup_workload(args.workload, args, m, tiling, 0, 1, 0, None, sqrtspp=args.sqrtspp)
    touched = set()
    ref = None
    for v in args.variants:
        name, _, assigns = v.partition(":")
        for k in touched:
            os.environ.pop(k, None)
        for a in filter(None, assigns.split(",")):
            k, _, val = a.partition("=")
            os.environ[k] = val
            touched.add(k)
        try:
            secs, stats = bench.run_steps(wl, args.steps, 1, 1, None)
        except Exception as e:  # a variant the library refuses
            print(json.dumps(dict(variant=name, error=str(e))), flush=True)
            continue
        ms = [s["kernel_ms"] for s in stats]
        rays = stats[-1]["rays"]
        frame = wl.tile.cpu().numpy()
        checksum = float(frame.sum())
        same = None
        if ref is None:
            ref = frame.copy()
        else:
            same = bool(np.array_equal(ref, frame))
        line = dict(variant=name, env=assigns, ms_best=round(min(ms), 3), ms_mean=round(sum(ms) / len(ms), 3),
                    Mray_s=round(rays / min(ms) / 1e3, 1), rays=rays, kernel_id=stats[-1].get("kernel_id"),
                    searches=stats[-1].get("knn_searches"), checksum=checksum, same_bits_as_first=same)
        if stats[-1].get("node_tests"):  # MCRT_COUNT_TESTS=1
            line["node_tests_per_ray"] = round(stats[-1]["node_tests"] / rays, 3)
            line["prim_tests_per_ray"] = round(stats[-1]["prim_tests"] / rays, 3)
        if stats[-1].get("knn_searches"):
            line["Msearch_s"] = round(stats[-1]["knn_searches"] / min(ms) / 1e3, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
