#!/usr/bin/env python3
"""bench.py — headline benchmark: Mray/s of the path-tracing hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c2_ggx|c1] [--no-cpu]

One "step" = one full frame of the workload: every owned pixel traced with spp samples by the HIP
integrator kernel (ray generation, BVH traversal, shading, NEE shadow rays, film accumulation), the
image rows left in HBM; with N > 1 the framebuffer rows are dealt to the ranks in groups of 8 and each
step ends with ONE RCCL gather of the packed rows to rank 0 over xGMI.

Workload (BASELINE.json configs[1], the config the metric is quoted on): hexagon_room.json camera 0,
1920x1080 @ 256 spp, scene image tests/golden/hexagon_room.mcrt (flattened by the reference's own
loader/BVH builder; synthetic = no external data needed). Inputs (scene arrays, Sobol tables) are
resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     algorithmic bytes (B_ray = n_node*64 + n_tri*72 + n_sphere*32 + 300,
               per-ray counts measured by the reference-equivalent oracle) per launch / kernel time
               measured with HIP events on the kernel's stream, against the 8 TB/s HBM peak;
  cpu_baseline the CPU integrator timed on this box's host cores on a bounded sample of the SAME
               workload (rows of the same frame at the same spp).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (scene image, width, height, sqrtspp, description)
    "c2": ("hexagon_room.mcrt", 1920, 1080, 16, "hexagon_room.json cam0 1920x1080 @ 256 spp (BASELINE configs[1])"),
    "c2_ggx": ("hexagon_room_ggx.mcrt", 1920, 1080, 16, "hexagon_room.json + GGX roughness, 1920x1080 @ 256 spp"),
    "c1": ("hexagon_room_diffuse.mcrt", 256, 256, 2, "hexagon_room_diffuse.json 256x256 @ 4 spp (BASELINE configs[0])"),
    # secondary: photon-mapped frame (BASELINE configs[4] class of work on the scene that is available):
    # 1e7 photon paths emitted on the GPU, maps built on the host, then timed eye passes with kNN estimates
    "pm": ("hexagon_room_pm.mcrt", 1920, 1080, 2, "hexagon_room.json photon mapping: 1e6 emissions x caustic_factor 10, k=50, 1920x1080 @ 4 spp"),
    # secondary (not the headline): a real BVH that does not fit in LDS; image made by tests/large/make_large.py
    "spaceship": ("../../oracle/_ref/images/spaceship.mcrt", 1920, 1080, 8,
                  "spaceship.json (68 760 of 457 200 triangles present), quaternary SAH, 1920x1080 @ 64 spp"),
    # BASELINE configs[2] at full size; the Stanford bunny is not in the reference tree (.MISSING_LARGE_BLOBS), a
    # synthetic 81 920-triangle stand-in is (tests/large/make_synthetic.py). The 120 MB image is flattened on this
    # machine by the reference's loader + BVH builder (tests/large/make_large.py:ensure_c3_image)
    "c3": ("../../oracle/_ref/images/metal_bunnies_c3.mcrt", 1920, 1080, 32,
           "metal_bunnies.json (stand-in bunny mesh, 491 592 triangles), quaternary SAH, 1920x1080 @ 1024 spp (BASELINE configs[2])"),
    # BASELINE configs[3] on ONE GPU (the 8-GPU figure is the driver's scaling run): the two missing hull meshes replaced by
    # stand-ins of the same triangle counts (tests/large/gen_mesh.c), 457 200 triangles in total
    "c4": ("../../oracle/_ref/images/spaceship_c4.mcrt", 3840, 2160, 32,
           "spaceship.json (stand-in hull meshes, 457 200 triangles), quaternary SAH, 3840x2160 @ 1024 spp (BASELINE configs[3])"),
    # BASELINE configs[4]: water.obj replaced by a 6 734 450-triangle heightfield; photons emitted on the GPU
    # (--emissions x caustic_factor 10 paths), octrees built on the host, timed eye passes with k = 50 estimates
    "c5": ("../../oracle/_ref/images/water_caustics_c5.mcrt", 1000, 1000, 16,
           "water_caustics.json (stand-in water surface, 6 898 815 triangles), octree BVH, photon map, 1000x1000 @ 256 spp (BASELINE configs[4])"),
}
# reference-side scene + flags for the cpu_baseline "reference" leg
REF_SCENES = {
    "hexagon_room.mcrt": ("hexagon_room.json", []),
    "metal_bunnies_c3.mcrt": ("metal_bunnies.json", ["--bvh", "quaternary_sah", "--bins", "8"]),
    "spaceship_c4.mcrt": ("spaceship.json", []),
}
SEED = 0x12345678
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHARD_ROWS = 8


def cpu_baseline(m, img, cam, budget_s=15.0, integrator=0, pm_maps=None):
    """Times the CPU integrator on rows of the same frame. Prefers the reference itself
    (oracle/_ref/mcrt_ref + oracle/_ref/scenes, both produced by oracle/Makefile in the build
    container); otherwise the C restatement (oracle/, kind "port"). Also returns the per-ray
    node/primitive test counts of the reference-equivalent traversal (for the roofline)."""
    import oracle_lib  # checker, cpu_baseline leg only

    threads = oracle_lib.hardware_threads()
    if pm_maps is not None:
        class _WithMaps:  # the scene image with the photon maps of this run (same maps the GPU uses)
            scene = img.scene
            path = img.path

            @staticmethod
            def photons(which):
                return pm_maps[which].desc

            @staticmethod
            def param(key):
                return {"k_nearest_photons": 50, "direct_visualization": 0}.get(key, 0)
        img = _WithMaps
    mid = cam.height // 2
    # calibration: 2 rows
    t0 = time.time()
    _, info = oracle_lib.render(img, cam, SEED, integrator, rows=(mid, mid + 2), threads=threads)
    per_row = max(info["seconds"] / 2.0, 1e-4)
    rows = int(max(2, min(cam.height, budget_s / per_row)))
    r0 = max(0, mid - rows // 2)
    _, info = oracle_lib.render(img, cam, SEED, integrator, rows=(r0, r0 + rows), threads=threads)
    rays = info["rays"]
    counts = dict(rays=rays, paths=info["paths"], node_per_ray=info["node_tests"] / rays,
                  tri_per_ray=(info["prim_tests"] - info["sphere_tests"]) / rays,
                  sphere_per_ray=info["sphere_tests"] / rays, rays_per_path=rays / info["paths"])
    if info["knn_searches"]:
        counts["knn_searches_per_s"] = info["knn_searches"] / info["seconds"]
        counts["knn_octants_per_search"] = info["knn_octants"] / info["knn_searches"]
        counts["knn_photons_per_search"] = info["knn_photons"] / info["knn_searches"]
    port = dict(value=rays / info["seconds"] / 1e6, unit="Mray/s", cores=threads, kind="port",
                sample="rows %d-%d of %dx%d @ %d spp (%d paths, %.1f s)" % (r0, r0 + rows, cam.width, cam.height, cam.sqrtspp ** 2,
                                                                          info["paths"], info["seconds"]))
    if "knn_searches_per_s" in counts:
        port["knn_searches_per_s"] = counts["knn_searches_per_s"]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
    ref_name, ref_flags = REF_SCENES.get(os.path.basename(img.path), (None, []))
    ref_scene = os.path.join(ROOT, "oracle", "_ref", "scenes", ref_name or "-")
    base = port
    if os.path.exists(ref_bin) and ref_name and os.path.exists(ref_scene):
        try:
            out = subprocess.run([ref_bin, "render", "--scene", ref_scene] + ref_flags + ["--width", str(cam.width), "--height", str(cam.height),
                                  "--sqrtspp", str(cam.sqrtspp), "--rows", str(r0), str(r0 + rows), "--out-radiance", "/dev/null"],
                                 capture_output=True, text=True, timeout=600, env=dict(os.environ, MCRT_REF_SEED=str(SEED)))
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            r = json.loads(line)
            base = dict(value=rays / r["seconds"] / 1e6, unit="Mray/s", cores=r["threads"], kind="reference",
                        sample="reference Camera::samplePixel, rows %d-%d of %dx%d @ %d spp (%d paths, %.1f s)"
                               % (r0, r0 + rows, cam.width, cam.height, cam.sqrtspp ** 2, r["paths"], r["seconds"]),
                        port_value=port["value"])
        except Exception as ex:  # keep the port numbers
            base = dict(port, note="reference run failed: %r" % (ex,))
    return base, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (roofline then uses stored counts)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--host-octree", action="store_true", help="build the photon octrees with the host builder instead of the GPU-assisted one")
    ap.add_argument("--emissions", type=float, default=1e6, help="photon_map.emissions of the photon-mapped workloads (x caustic_factor 10 paths)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MCRT_BENCH_SHARE_GPU=1 (rehearsal of the N > 1 path on a one-GPU box, not a measurement): every rank uses cuda:0 and the
    # collectives run over gloo, which moves CUDA tensors through the host
    share_gpu = world > 1 and os.environ.get("MCRT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    image_file, W, H, sqrtspp, desc = WORKLOADS[args.workload]
    if args.workload in ("c3", "c4", "c5"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "large"))
        import make_large
        if local_rank == 0 and make_large.ensure_image(args.workload) is None:
            raise SystemExit("%s needs oracle/_ref (python __graft_entry__.py build in the build container)" % args.workload)
        if world > 1:
            dist.barrier()
    img = m.SceneImage(os.path.join(ROOT, "tests", "golden", image_file))
    cam = img.camera
    cam.width, cam.height, cam.sqrtspp = W, H, sqrtspp
    full = cam.copy()
    cam = tiling.shard_camera(full, rank, world, SHARD_ROWS)
    ctx = m.Context(local_rank)
    ctx.upload_image(img)  # scene resident in HBM before the timed region
    integrator = m.INTEGRATOR_PATH_TRACER
    pm_maps = None
    emit_info = None
    photon_workload = args.workload in ("pm", "c5")
    if photon_workload:
        # emission pass on the GPU (sharded over the ranks and all-gathered), octrees on the host, upload
        integrator = m.INTEGRATOR_PHOTON_MAPPER
        em = ctx.emit_photons(args.emissions, 10.0, SEED, rank, world)
        emit_info = dict(paths=em["paths"], rays=em["rays"], kernel_ms=em["kernel_ms"])
        lists = []
        for name in ("global_", "caustic"):
            ph = torch.from_numpy(em[name][0]).to(torch.device("cuda", local_rank))
            if world > 1:
                sizes = [torch.zeros(1, dtype=torch.int64, device=ph.device) for _ in range(world)]
                dist.all_gather(sizes, torch.tensor([ph.shape[0]], dtype=torch.int64, device=ph.device))
                cap = int(max(int(x.item()) for x in sizes))
                pad = torch.zeros((cap, 8), dtype=torch.float32, device=ph.device)
                pad[: ph.shape[0]] = ph
                parts = [torch.empty_like(pad) for _ in range(world)]
                dist.all_gather(parts, pad)
                ph = torch.cat([parts[r][: int(sizes[r].item())] for r in range(world)])
            lists.append(ph.cpu().numpy())
        sc = img.scene
        t_build = time.perf_counter()
        # octrees: cell codes + radix sort + gather + leaf boxes on the GPU, octant assembly on the host (--host-octree: all on the host)
        bctx = None if args.host_octree else ctx
        pm_maps = (m.PhotonMap(lists[0], sc.bb_min[:], sc.bb_max[:], 200, ctx=bctx), m.PhotonMap(lists[1], sc.bb_min[:], sc.bb_max[:], 200, ctx=bctx))
        emit_info.update(global_photons=int(lists[0].shape[0]), caustic_photons=int(lists[1].shape[0]),
                         octree_build_s=time.perf_counter() - t_build, octree_builder="host" if args.host_octree else "gpu",
                         emission_Mray_per_s=emit_info["rays"] / max(emit_info["kernel_ms"], 1e-9) / 1e3)
        ctx.upload_photons(pm_maps[0].desc, pm_maps[1].desc, 50, False)

    my_rows = m.shard_rows(cam)
    max_rows = tiling.max_rows(full, world, SHARD_ROWS)
    dev = torch.device("cuda", local_rank)
    tile = torch.zeros((max_rows, W, 3), dtype=torch.float64, device=dev)  # packed owned rows (+ padding)
    gathered = [torch.empty_like(tile) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream(dev).cuda_stream

    stats_acc = []

    def step():
        ctx.render_device(cam, SEED, integrator, tile.data_ptr(), stream)
        st = ctx.render_finish()
        if world > 1:
            dist.gather(tile, gathered, dst=0)  # the single collective of the data path
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats_acc.append(step())
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(sum(s["rays"] for s in stats_acc)), float(sum(s["paths"] for s in stats_acc)),
                         float(sum(s["kernel_ms"] for s in stats_acc)), float(sum(s["knn_searches"] for s in stats_acc))],
                        dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kmax = rays[2:3].clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        rays[2] = kmax[0]
    elapsed = float(t.item())
    total_rays, total_paths, kernel_ms_sum, total_knn = float(rays[0]), float(rays[1]), float(rays[2]), float(rays[3])

    if rank == 0:
        # sanity: the gathered frame is complete and finite
        if world > 1:
            frame = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
            for r in range(world):
                rows = torch.from_numpy(tiling.rows_of(full, r, world, SHARD_ROWS)).to(dev)
                frame[rows] = gathered[r][: len(rows)]
        else:
            frame = tile[: len(my_rows)]
        finite = bool(torch.isfinite(frame).all().item())
        mean = float(frame.mean().item())

        result = {
            "metric": "Mray/s (whole node), 1920x1080 @ 256 spp path trace" if args.workload.startswith("c2") else "Mray/s (whole node)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": desc, "width": W, "height": H, "spp": sqrtspp ** 2, "integrator": "photon_mapper" if photon_workload else "path_tracer",
                       "seed": SEED, "sharding": "rows in groups of %d, round-robin over %d GPU(s)" % (SHARD_ROWS, world),
                       "rays_per_step": total_rays / args.steps, "paths_per_step": total_paths / args.steps,
                       "frame_mean_radiance": mean, "frame_finite": finite,
                       "knn_searches_per_s": total_knn / elapsed if total_knn else None,
                       "photon_pass": emit_info if photon_workload else None},
        }
        counts = None
        if world == 1 and not args.no_cpu:
            base, counts = cpu_baseline(m, img, full, args.cpu_seconds, integrator, pm_maps)
            result["cpu_baseline"] = base
        if counts is None:
            # per-ray counts of the reference-equivalent traversal measured on this workload by the oracle
            # (DESIGN.md "Measurement"); used when the CPU leg is skipped (N > 1)
            counts = {"spaceship": dict(node_per_ray=33.44, tri_per_ray=6.96, sphere_per_ray=0.0),
                      "c3": dict(node_per_ray=46.28, tri_per_ray=7.17, sphere_per_ray=0.02),
                      "c4": dict(node_per_ray=71.99, tri_per_ray=14.22, sphere_per_ray=0.0),
                      "c5": dict(node_per_ray=45.05, tri_per_ray=7.28, sphere_per_ray=0.0),
                      "pm": dict(node_per_ray=14.34, tri_per_ray=9.30, sphere_per_ray=7.40),
                      }.get(args.workload, dict(node_per_ray=13.82, tri_per_ray=8.61, sphere_per_ray=6.31))
        b_ray = counts["node_per_ray"] * 64 + counts["tri_per_ray"] * 72 + counts["sphere_per_ray"] * 32 + 300
        launches = args.steps * 1  # one integrator launch per step per GPU
        kernel_ms = kernel_ms_sum / launches
        rays_per_launch = total_rays / args.steps / world
        achieved = rays_per_launch * b_ray / (kernel_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if world == 1 and args.workload in tj:
                traffic = tj[args.workload]["traffic_bytes"]
        except Exception:
            pass
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "note": "achieved = reference-equivalent algorithmic bytes (SURVEY.md 8(d)) / kernel time; "
                                      "traffic = measured HBM bytes per launch (rocprofv3 PMC, profiles/). Scenes that fit in LDS "
                                      "move almost nothing through HBM, so frac can exceed 1 for them.",
                              "kernel": {"pm": "renderKernelPM", "c5": "renderKernelPM", "spaceship": "renderKernelSM", "c3": "wfTraceKernel + wfShadeKernel (all launches of the frame)",
                                         "c4": "wfTraceKernel + wfShadeKernel (all launches of the frame)"}.get(args.workload, "renderKernel<path_tracer, flat>"),
                              "kernel_ms": kernel_ms,
                              "bytes_per_ray": b_ray, "rays_per_launch": rays_per_launch,
                              "per_ray_counts": {k: counts[k] for k in ("node_per_ray", "tri_per_ray", "sphere_per_ray")}}
        if counts.get("knn_photons_per_search"):
            # B_knn = n_octant*128 + n_photon_scanned*32 + k*32 per search (reference-equivalent counts)
            b_knn = counts["knn_octants_per_search"] * 128 + counts["knn_photons_per_search"] * 32 + 50 * 32
            knn_rate = total_knn / args.steps / world / (kernel_ms * 1e-3)
            result["roofline"]["knn"] = {"bytes_per_search": b_knn, "searches_per_s_in_kernel": knn_rate,
                                          "achieved_GBs_incl_knn": achieved + knn_rate * b_knn / 1e9}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
