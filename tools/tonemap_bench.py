#!/usr/bin/env python3
"""Times mcrt_tonemap_device (Image::save on the GPU) on a synthetic 1920x1080 frame resident in HBM, next to the oracle's
Image::save on one host core. Prints one JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    pkg = importlib.import_module("monte-carlo-ray-tracer_amd")
    import oracle_lib
    w, h = 1920, 1080
    rng = np.random.default_rng(1)
    rgb = rng.random((h, w, 3)) ** 3 * 8.0
    ctx = pkg.Context(0)
    frame = torch.from_numpy(rgb).to("cuda:0")
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    res = {}
    for tm in ("HABLE", "ACES"):
        d = pkg.ImageDesc.make(w, h, tm, False, -0.25, 0.0)
        for _ in range(3):
            ctx.tonemap_device(frame.data_ptr(), d, out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            factors = ctx.tonemap_device(frame.data_ptr(), d, out.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        want, wf = oracle_lib.image_save(rgb, pkg.TONEMAPPERS[tm], False, -0.25, 0.0)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        diff = np.count_nonzero(out.cpu().numpy() != want)
        res[tm] = dict(gpu_ms_per_frame=round(ms, 3), cpu_oracle_ms=round(cpu_ms, 1), factors_equal=factors == wf, bytes_differing=int(diff),
                       algorithmic_GBps=round((w * h * (5 * 24 + 3)) / (ms * 1e-3) / 1e9, 1))
    print(json.dumps(dict(frame="%dx%d" % (w, h), **res)))


if __name__ == "__main__":
    main()
