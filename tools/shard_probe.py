#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU, for the configurations BASELINE names: what one rank of an N-GPU job does - its shard of the
frame (rows in groups of 8, round robin: monte-carlo-ray-tracer_amd/tiling.py, exactly what bench.py --gpus N renders) - timed alone
for N = 1, 2, 4, 8, shard by shard. The N-GPU frame time is the SLOWEST shard (the barrier), so

    predicted speed-up(N) = t(1 shard of 1) / max over k of t(shard k of N)

and everything per-rank that does not shrink with N shows up in it: launch overheads of the wavefront pipeline (one shade + one trace
launch per bounce whatever the shard holds), the tail of every pass, the resolve. Not in it: the one gather at the end (50-200 MB over
xGMI: < 1 ms at link speed against frames of seconds) and, for photon-mapped frames, the all-gather of the photon lists (stated as
bytes; RCCL ring over 7 xGMI links).

Photon-mapped workloads (c5, pm) also time the photon pass as bench.py's N > 1 path runs it: THIS rank's shard of the emission paths
(mcrt_emit_photons_device), then both maps built from the FULL lists (mcrt_upload_photons_device) - the part every rank repeats.

    python tools/shard_probe.py <workload> [--sqrtspp S] [--emissions E] [--shards all|ends] [--n 1,2,4,8]

Prints one JSON object: per N the slowest shard's ms, the predicted speed-up and efficiency, per-shard times, photon-pass times."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--sqrtspp", type=int, default=None)
    ap.add_argument("--emissions", type=float, default=None)
    ap.add_argument("--shards", default="all", help="all: every shard of every N (max = the prediction); ends: first and last only")
    ap.add_argument("--n", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--host-octree", action="store_true")
    args = ap.parse_args()
    import bench
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    wl = bench.setup_workload(args.workload, args, m, tiling, 0, 1, 0, None, sqrtspp=args.sqrtspp)
    ns = [int(x) for x in args.n.split(",")]
    out = {"workload": wl.desc, "kernel": None, "N": {}}
    t1 = None
    for n in ns:
        per = []
        for index in (range(n) if args.shards == "all" else sorted({0, n - 1})):
            shard = tiling.shard_camera(wl.full, index, n, bench.SHARD_ROWS)
            rows = len(m.shard_rows(shard))
            buf = torch.zeros((max(rows, 1), wl.W, 3), dtype=torch.float64, device=wl.dev)
            best, launches = 1e30, 0
            for _ in range(args.reps):
                torch.cuda.synchronize(wl.dev)
                t0 = time.perf_counter()
                wl.ctx.render_device(shard, bench.SEED, wl.integrator, buf.data_ptr())
                st = wl.ctx.render_finish()
                torch.cuda.synchronize(wl.dev)
                best = min(best, (time.perf_counter() - t0) * 1e3)  # wall time of the shard: launches, host checks and all
                launches = st["kernel_launches"]
                out["kernel"] = m.KERNEL_NAMES.get(st["kernel_id"], "?")
            per.append(dict(shard=index, rows=rows, ms=round(best, 2), kernel_ms=round(st["kernel_ms"], 2), launches=launches, rays=st["rays"]))
            del buf
        worst = max(p["ms"] for p in per)
        t1 = t1 or worst
        out["N"][n] = dict(slowest_shard_ms=round(worst, 2), predicted_speedup=round(t1 / worst, 3), efficiency=round(t1 / (n * worst), 3),
                           gather_MB=round(wl.W * wl.H * 24 / 1e6, 1), shards=per)
    if wl.photon:
        sc = wl.img.scene
        pp = {}
        for n in ns:
            best_emit, em = 1e30, None
            for _ in range(args.reps):
                torch.cuda.synchronize(wl.dev)
                t0 = time.perf_counter()
                em = wl.ctx.emit_photons_device(wl.emissions, 10.0, bench.SEED, n - 1, n)  # the last shard: lights are dealt in order
                torch.cuda.synchronize(wl.dev)
                best_emit = min(best_emit, (time.perf_counter() - t0) * 1e3)
            pp[n] = dict(emit_shard_ms=round(best_emit, 2), emit_kernel_ms=round(em["kernel_ms"], 2),
                         shard_photons=[int(em["global_"][1]), int(em["caustic"][1])])
        # the full lists once (N = 1 emission), then the map build every rank repeats, warm (work buffers pooled in the context)
        em = wl.ctx.emit_photons_device(wl.emissions, 10.0, bench.SEED, 0, 1)

        class _Dev:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n, 8), "typestr": "<f4", "data": (ptr, False), "version": 2}
        lists = [torch.as_tensor(_Dev(*em[k]), device=wl.dev).clone() for k in ("global_", "caustic")]
        torch.cuda.synchronize(wl.dev)
        build = []
        for _ in range(args.reps + 1):
            t0 = time.perf_counter()
            wl.ctx.upload_photons_device(lists[0].data_ptr(), lists[0].shape[0], lists[1].data_ptr(), lists[1].shape[0], sc.bb_min[:], sc.bb_max[:], 200, 50, False)
            torch.cuda.synchronize(wl.dev)
            build.append((time.perf_counter() - t0) * 1e3)
        list_bytes = sum(int(x.shape[0]) * 32 for x in lists)
        for n in ns:
            # ring all-gather: every rank receives (n - 1) / n of the lists; 7 xGMI links x ~153 GB/s per GPU, a ring uses one in each direction
            est_ms = 0.0 if n == 1 else list_bytes * (n - 1) / n / 153e9 * 1e3
            pp[n].update(map_build_ms=round(min(build[1:]), 2), map_build_first_ms=round(build[0], 2), allgather_bytes=list_bytes,
                         allgather_ms_at_one_link=round(est_ms, 2),
                         photon_pass_ms_per_rank=round(pp[n]["emit_shard_ms"] + min(build[1:]) + est_ms, 2))
            o = out["N"][n]
            o["frame_with_photon_pass_ms"] = round(o["slowest_shard_ms"] + pp[n]["photon_pass_ms_per_rank"], 2)
        f1 = out["N"][ns[0]]["frame_with_photon_pass_ms"]
        for n in ns:
            out["N"][n]["predicted_speedup_with_photon_pass"] = round(f1 / out["N"][n]["frame_with_photon_pass_ms"], 3)
        out["photon_pass"] = pp
    print(json.dumps(out))
    wl.ctx.close()


if __name__ == "__main__":
    main()
