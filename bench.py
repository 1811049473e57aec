#!/usr/bin/env python3
"""bench.py — headline benchmark: Mray/s of the path-tracing hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|...] [--no-cpu] [--no-secondary] [--no-counters]

One "step" = one full frame of the workload: every owned pixel traced with spp samples by the HIP integrator kernel(s)
(ray generation, BVH traversal, shading, NEE shadow rays, film accumulation), the image rows left in HBM; with N > 1 the
framebuffer rows are dealt to the ranks in groups of 8 and each step ends with ONE RCCL gather of the packed rows to
rank 0 over xGMI.

Headline workload (BASELINE.json configs[1], the config the metric is quoted on): hexagon_room.json camera 0,
1920x1080 @ 256 spp, scene image tests/golden/hexagon_room.mcrt (flattened by the reference's own loader/BVH builder).
Inputs (scene arrays, Sobol tables, photon maps) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line (contract in the task statement). At N = 1 it also carries
  roofline      of the frame's kernels, the SAME block for the headline and every secondary leg (leg_roofline): three fractions of
                physical peaks - `hbm_frac_measured` (fabric bytes of one frame from rocprofv3 PMC passes made inside this run - child
                processes of this script - / kernel time / 8 TB/s), `valu_issue_frac` (SQ_THREAD_CYCLES_VALU / kernel time / 1024 SIMDs x
                16 lanes x 2.4 GHz = VALU busy x lane utilisation) and `frac_necessary` (FP64 lane-ops of the REFERENCE's algorithm on the
                same rays / that peak); `frac` = the larger of the two measured ones, `bound` names it. `algorithmic_GBs` /
                `algorithmic_frac` = reference-equivalent bytes (B_ray = n_node*64 + n_tri*72 + n_sphere*32 + 300,
                per-ray counts of the reference's best-first traversal measured by the oracle) per second of kernel time - a work
                rate that caches serve, reported beside the bound and never as `frac`; `traffic` = the measured fabric bytes of a frame
                (integrator kernels + sampleResolveKernel);
  cpu_baseline  the reference's own integrator (oracle/_ref/mcrt_ref, kind "reference") on this box's host cores, on rows of the
                SAME frame, at all hardware threads and at the thread count where it runs best (its shared_ptr reference counts
                contend: `best_value` at `best_cores` is the baseline to quote), and the C restatement ("port_value");
  parity        c2 and c2_ggx: the frame's rows 536-540 against the reference's radiance for those rows (tests/golden, made by the
                reference itself): max relative error, the number of pixels beyond 1e-4, bit_identical;
  secondary     driver-timed legs: c2_ggx (the "GGX + Fresnel" reading of BASELINE configs[1]), spaceship cockpit (renderKernelSM),
                photon-mapped hexagon_room (renderKernelPM, kNN), metal_bunnies C3 and spaceship C4 (wavefront pipeline), water_caustics
                C5 (photon mapper; `frame_with_photon_pass_ms` adds the device photon pass of 1e8 emission paths to the eye pass) -
                each with its own roofline block and cpu_baseline.
"""
import argparse
import glob
import importlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
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
    # photon-mapped frame (BASELINE configs[4] class of work on the scene that is available): photon paths emitted on the
    # GPU, maps built GPU-assisted, then timed eye passes with kNN estimates
    "pm": ("hexagon_room_pm.mcrt", 1920, 1080, 2, "hexagon_room.json photon mapping: 1e6 emissions x caustic_factor 10, k=50, 1920x1080 @ 4 spp"),
    # a real BVH that does not fit in LDS; image made by integration/large_scenes/make_large.py
    "spaceship": ("../../oracle/_ref/images/spaceship.mcrt", 1920, 1080, 8,
                  "spaceship.json (68 760 of 457 200 triangles present), quaternary SAH, 1920x1080 @ 64 spp"),
    # BASELINE configs[2] at full size; the Stanford bunny is not in the reference tree (.MISSING_LARGE_BLOBS), a
    # synthetic 81 920-triangle stand-in is (integration/large_scenes/make_synthetic.py). The 120 MB image is flattened on this
    # machine by the reference's loader + BVH builder (integration/large_scenes/make_large.py:ensure_image)
    "c3": ("../../oracle/_ref/images/metal_bunnies_c3.mcrt", 1920, 1080, 32,
           "metal_bunnies.json (stand-in bunny mesh, 491 592 triangles), quaternary SAH, 1920x1080 @ 1024 spp (BASELINE configs[2])"),
    # BASELINE configs[3] on ONE GPU: the two missing hull meshes replaced by stand-ins of the same triangle counts
    "c4": ("../../oracle/_ref/images/spaceship_c4.mcrt", 3840, 2160, 32,
           "spaceship.json (stand-in hull meshes, 457 200 triangles), quaternary SAH, 3840x2160 @ 1024 spp (BASELINE configs[3])"),
    # BASELINE configs[4]: water.obj replaced by a 6 734 450-triangle heightfield; photons emitted on the GPU
    "c5": ("../../oracle/_ref/images/water_caustics_c5.mcrt", 1000, 1000, 16,
           "water_caustics.json (stand-in water surface, 6 898 815 triangles), octree BVH, photon map, 1000x1000 @ 256 spp (BASELINE configs[4])"),
}
# reference-side scene + flags for the cpu_baseline "reference" leg (scene copies under oracle/_ref/scenes, made by
# integration/large_scenes/make_large.py in the build container; they travel to the GPU box with the snapshot)
REF_SCENES = {
    "hexagon_room.mcrt": ("hexagon_room.json", []),
    "hexagon_room_ggx.mcrt": ("hexagon_room.json", ["--specular-roughness", "green", "0.1", "--specular-roughness", "crystal", "0.05"]),
    "metal_bunnies_c3.mcrt": ("metal_bunnies.json", ["--bvh", "quaternary_sah", "--bins", "8"]),
    "spaceship_c4.mcrt": ("spaceship.json", []),
    "spaceship.mcrt": ("spaceship_cockpit.json", []),
    "hexagon_room_pm.mcrt": ("hexagon_room.json", []),
    "water_caustics_c5.mcrt": ("water_caustics.json", []),
}
# c2_ggx: the "GGX + Fresnel" reading of BASELINE configs[1] (hexagon_room.json with specular_roughness 0.1 on
# `green` and 0.05 on `crystal`, "report both"): same frame size as the headline, its own parity block against reference-made rows
SECONDARY = ("c2_ggx", "spaceship", "pm", "c3", "c4", "c5")
# legs whose scene is the reference's own file as shipped (the others replace meshes the reference tree lacks, .MISSING_LARGE_BLOBS,
# by deterministic stand-ins of the same triangle counts): what the line's "data" says
REFERENCE_SCENE_LEGS = ("c1", "c2", "c2_ggx", "pm")
# per leg: timed steps (None = --secondary-steps), spp of the untimed warm-up frame (None = the leg's own), spp of the frame the PMC
# child passes count (None = the leg's own; per-sample work is the same at any spp, the scale is stated in frame_scale)
LEG_PLAN = {"c3": dict(steps=2, warm_sqrtspp=None, pmc_sqrtspp=8),   # 5.2 s a frame: two timed frames after a warm-up frame of the SAME size (a smaller
                                                                  # warm-up frame leaves the pool, queue and sample store to be allocated inside the first timed
                                                                  # frame: measured +0.55 s per frame over two frames, round 6)
            "c4": dict(steps=2, warm_sqrtspp=4, pmc_sqrtspp=4),   # a 4K @ 1024 spp frame is ~29 s: two timed frames after a 16 spp warm-up frame (the first one
                                                                  # pays the allocations: ~1 % of the pair)
            "c5": dict(steps=None, warm_sqrtspp=None, pmc_sqrtspp=6)}  # (36 spp = 36 M path samples: the counted frame must take the path the timed
                                                                        # one takes - photon-mapped frames go through the pipeline from 32 M, csrc/mcrt_hip.hip)
# photon_map.emissions of the photon-mapped workloads (x caustic_factor 10 paths): BASELINE configs[4] says 1e8 emission paths for C5
EMISSIONS = {"pm": 1e6, "c5": 1e7}
# emissions of the REFERENCE's own photon pass in the cpu_baseline leg of C5 (its CPU emission pass at 1e8 paths takes minutes; the
# timed part is its eye pass only, and paths / rays / searches per row do not depend on the map)
REF_EMISSIONS = {"pm": 1e6, "c5": 1e6}
SEED = 0x12345678
# --rehearse-dist: the collectives of the N > 1 path (RCCL init, the photon all-gathers, the per-frame gather, barriers, reductions,
# destroy) run with a process group of ONE rank on one GPU, so that branch has executed before the driver's only 8-GPU run
REHEARSE_DIST = False


def _collectives(world):
    return world > 1 or REHEARSE_DIST

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# vector-ALU peak in FP64-rate lane slots: 256 CUs x 4 SIMDs, a wave64 FP64 instruction occupies its SIMD for 4 cycles
# (16 lanes per cycle), 2.4 GHz max clock (MI355X_MICROARCH.md chip parameters) = 39.3e12 lane-ops/s (x2 = 78.6 TFLOP/s FMA)
VALU_SIMDS = 1024
VALU_PEAK_GLANEOPS = VALU_SIMDS * 16 * 2.4
SHARD_ROWS = 8
# stored per-ray counts of the reference-equivalent traversal (oracle, DESIGN.md "Measurement"): used when the CPU leg is skipped
STORED_COUNTS = {"spaceship": dict(node_per_ray=33.44, tri_per_ray=6.96, sphere_per_ray=0.0),
                 "c3": dict(node_per_ray=46.28, tri_per_ray=7.17, sphere_per_ray=0.02),
                 "c4": dict(node_per_ray=71.99, tri_per_ray=14.22, sphere_per_ray=0.0),
                 "c5": dict(node_per_ray=45.05, tri_per_ray=7.28, sphere_per_ray=0.0),
                 "pm": dict(node_per_ray=14.34, tri_per_ray=9.30, sphere_per_ray=7.40)}
DEFAULT_COUNTS = dict(node_per_ray=13.82, tri_per_ray=8.61, sphere_per_ray=6.31)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (checker code: oracle/ and oracle/_ref are used here and nowhere in the timed region)
# ------------------------------------------------------------------------------------------------------------------
def _run_reference(img_path, cam, r0, r1, threads, photon_emissions=None, timeout=600):
    """Rows [r0, r1) of the frame by the reference binary; returns its JSON record or None. With photon_emissions the reference
    runs its own photon pass first (PhotonMapper's constructor, untimed) and the record's seconds cover its eye pass only."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
    ref_name, ref_flags = REF_SCENES.get(os.path.basename(img_path), (None, []))
    ref_scene = os.path.join(ROOT, "oracle", "_ref", "scenes", ref_name or "-")
    if not (os.path.exists(ref_bin) and ref_name and os.path.exists(ref_scene)):
        return None
    cmd = [ref_bin, "render", "--scene", ref_scene] + ref_flags + ["--width", str(cam.width), "--height", str(cam.height), "--sqrtspp", str(cam.sqrtspp),
                                                                   "--rows", str(r0), str(r1), "--out-radiance", "/dev/null"]
    if threads:
        cmd += ["--threads", str(threads)]
    if photon_emissions:
        cmd += ["--photon", "--emissions", str(int(photon_emissions))]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, MCRT_REF_SEED=str(SEED)))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def cpu_baseline(m, img, cam, budget_s=12.0, integrator=0, pm_maps=None, scan_threads=True, ref_threads=None, ref_emissions=None):
    """Times the CPU integrator on rows of the same frame. Prefers the reference itself (oracle/_ref/mcrt_ref +
    oracle/_ref/scenes, produced by oracle/Makefile in the build container); otherwise the C restatement (oracle/, kind
    "port"). Also returns the per-ray node/primitive test counts of the reference-equivalent traversal (for the
    algorithmic-bytes figure)."""
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
    # the port, calibrated on 2 rows
    _, info = oracle_lib.render(img, cam, SEED, integrator, rows=(mid, mid + 2), threads=threads)
    per_row = max(info["seconds"] / 2.0, 1e-4)
    rows = int(max(2, min(cam.height, 0.5 * budget_s / per_row)))
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
    spp = cam.sqrtspp ** 2
    port = dict(value=rays / info["seconds"] / 1e6, unit="Mray/s", cores=threads, kind="port",
                sample="rows %d-%d of %dx%d @ %d spp (%d paths, %.1f s)" % (r0, r0 + rows, cam.width, cam.height, spp, info["paths"], info["seconds"]))
    if "knn_searches_per_s" in counts:
        port["knn_searches_per_s"] = counts["knn_searches_per_s"]
    base = port
    if integrator == 0:
        try:
            rays_per_row = rays / rows  # rays of a row: the port traces the reference's paths (bit-identical frames)
            # the reference at all hardware threads on ONE row, then (headline only) at fewer threads: its BVH::intersect
            # copies shared_ptrs, whose reference counts contend, so all threads is rarely its best
            cal = {}
            for t in ([threads] + ([t for t in (64, 32, 16) if t < threads] if scan_threads else [])):
                r = _run_reference(img.path, cam, mid, mid + 1, t)
                if r is None:
                    break
                cal[t] = rays_per_row / r["seconds"]
            if ref_threads and ref_threads not in cal and cal:
                r = _run_reference(img.path, cam, mid, mid + 1, ref_threads)
                if r:
                    cal[ref_threads] = rays_per_row / r["seconds"]
            if cal:
                best_t = max(cal, key=cal.get) if (scan_threads or not ref_threads) else ref_threads

                def sample_at(t, seconds):  # rows around the middle of the frame worth `seconds` of the reference at t threads
                    n = int(max(1, min(cam.height, seconds * cal[t] / rays_per_row)))
                    q = min(max(0, mid - n // 2), cam.height - n)
                    return _run_reference(img.path, cam, q, q + n, t), q, n

                if scan_threads:
                    # the headline's baseline, as BASELINE.md section 3 plans it: the reference with EVERY hardware thread of this host
                    # (num_render_threads = -1, integrator.cpp:20-24) on >= 25 s of its own work. Beside it the thread count where the
                    # reference runs best on this host (best_value / best_cores): BVH::intersect copies a shared_ptr per visited node
                    # (bvh.cpp:80-129) whose reference counts contend across sockets, so fewer threads are faster - an artefact of the
                    # reference's host code, quoted, not chosen.
                    r, q0, n_rows = sample_at(threads, max(25.0, budget_s))
                    rb, _, nb = sample_at(best_t, 0.5 * budget_s) if best_t != threads else (r, q0, n_rows)
                    base = dict(value=rays_per_row * n_rows / r["seconds"] / 1e6, unit="Mray/s", cores=threads, kind="reference",
                                sample="reference Camera::samplePixel at all %d hardware threads: rows %d-%d of %dx%d @ %d spp (%d paths, %.1f s of CPU work); "
                                       "thread scan on row %d" % (threads, q0, q0 + n_rows, cam.width, cam.height, spp, r["paths"], r["seconds"], mid),
                                quoted="value = the reference at all hardware threads (BASELINE.md 3); best_value = at the thread count where it runs best on this host",
                                best_value=rays_per_row * nb / rb["seconds"] / 1e6, best_cores=best_t, best_sample_s=rb["seconds"],
                                all_threads_value=rays_per_row * n_rows / r["seconds"] / 1e6, all_threads_cores=threads,
                                by_threads={str(k): v / 1e6 for k, v in sorted(cal.items())},
                                port_value=port["value"], port_cores=threads)
                else:
                    # secondary legs: one sample at the thread count the headline found best (a few seconds each, so that the run stays short)
                    r, q0, n_rows = sample_at(best_t, 0.5 * budget_s)
                    base = dict(value=rays_per_row * n_rows / r["seconds"] / 1e6, unit="Mray/s", cores=best_t, kind="reference",
                                sample="reference Camera::samplePixel at %d of %d hardware threads (the headline's best count): rows %d-%d of %dx%d @ %d spp (%d paths, %.1f s)"
                                       % (best_t, threads, q0, q0 + n_rows, cam.width, cam.height, spp, r["paths"], r["seconds"]),
                                best_value=rays_per_row * n_rows / r["seconds"] / 1e6, best_cores=best_t,
                                all_threads_value=cal.get(threads, 0.0) / 1e6 or None, all_threads_cores=threads,
                                port_value=port["value"], port_cores=threads)
        except Exception as ex:  # keep the port numbers
            base = dict(port, note="reference run failed: %r" % (ex,))
    elif ref_emissions:
        # photon-mapped legs: the reference's own PhotonMapper::sampleRay (photon-mapper.cpp:279-391) on rows of the same frame. One
        # process = scene load + its CPU photon pass (both untimed) + the timed eye pass over the rows. Rays and searches of a row do
        # not depend on the photon map (estimates are only added up), so the port's per-row counts price the reference's rows too.
        try:
            rays_per_row, knn_per_row = rays / rows, info["knn_searches"] / rows
            t = ref_threads or threads
            n_rows = max(1, rows // 8)  # the port took ~budget/2 for `rows` rows; the reference is several times slower per row
            q0 = max(0, mid - n_rows // 2)
            r = _run_reference(img.path, cam, q0, q0 + n_rows, t, photon_emissions=ref_emissions, timeout=900)
            if r:
                base = dict(value=rays_per_row * n_rows / r["seconds"] / 1e6, unit="Mray/s", cores=t, kind="reference",
                            knn_searches_per_s=knn_per_row * n_rows / r["seconds"],
                            sample="reference PhotonMapper::sampleRay (eye pass only; its own photon pass of %.0e x 10 emission paths untimed), rows %d-%d of %dx%d @ %d spp at %d threads (%d paths, %.1f s)"
                                   % (ref_emissions, q0, q0 + n_rows, cam.width, cam.height, spp, t, r["paths"], r["seconds"]),
                            port_value=port["value"], port_cores=threads, port_knn_searches_per_s=port.get("knn_searches_per_s"))
        except Exception as ex:
            base = dict(port, note="reference run failed: %r" % (ex,))
    return base, counts


# ------------------------------------------------------------------------------------------------------------------
# in-run hardware counters: one frame of the workload in a child process under rocprofv3 --pmc (one pass per counter set)
# ------------------------------------------------------------------------------------------------------------------
SQ_SET = ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
          "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "GRBM_GUI_ACTIVE"]
# the typed VALU instruction counters (wave instructions by operand type; 8 SQ slots per pass: six ride with the write pass, whose SQ
# slots were free, two with the read pass): what the FP64 roofline fraction is made of (round 5)
SQ_TYPED = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
            "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32"]
# Two passes (TCC has 4 counter slots, SQ 8, GRBM 2 - MI355X_MICROARCH.md "rocprofv3 PMC slots"): reads by request size class
# together with the SQ set, then writes + L2 hit/miss. Calibrated on known byte counts in the kernels' own access patterns
# (tools/calibrate_traffic.py -> profiles/r03_traffic_calibration.json): every fabric read is a 128-byte L2 line, so
# read bytes = 32 n32 + 64 n64 + 128 n128 = 2 x FETCH_SIZE for streams, 8 KB photon runs, 64-byte blocks and 80-byte records alike
# (a scattered 64-byte block MOVES 128 bytes); WRITE_SIZE is exact for streamed stores and counts 32-byte sectors for scattered ones.
PMC_PASSES = (("read+sq", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"] + SQ_SET),
              ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"] + SQ_TYPED))
PMC_FALLBACK = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", SQ_SET), ("typed", SQ_TYPED))  # separate passes, if a combined pass is refused
# every kernel between the HIP events that time a frame: the integrator kernels and the in-order resolve of the per-sample store
INTEGRATOR_KERNELS = ("renderKernel", "wfTraceKernel", "wfShadeKernel", "wfKnnKernel", "sampleResolveKernel")
PROFILE_DIR = os.path.join(ROOT, "gpurun_out", "bench_profiles")  # per-leg counter summaries of THIS run (copied to profiles/ when committed)


def kernel_instance_name(kname):
    """`void (anonymous namespace)::renderKernelFlatK<768>((anonymous namespace)::DeviceScene, ...)` -> `renderKernelFlatK<768>`:
    the function's name with its template arguments, without return type, namespaces and parameter list."""
    k = kname.replace("(anonymous namespace)::", "").replace("mcrt::", "")
    if k.startswith("void "):
        k = k[5:]
    depth = 0
    for i, ch in enumerate(k):  # cut at the parameter list: the first '(' outside the template argument list
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            k = k[:i]
            break
    return k.strip()


def _pmc_pass(rocprof, tag, names, workload, sqrtspp, emissions, timeout, out):
    """One rocprofv3 --pmc pass over one frame (child process). Returns True when counters came back."""
    tmp = tempfile.mkdtemp(prefix="mcrt_pmc_", dir="/tmp")
    got = False
    try:
        cmd = [rocprof, "--kernel-trace", "--pmc"] + names + ["-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--child-frame", "--workload", workload]
        if sqrtspp:
            cmd += ["--sqrtspp", str(sqrtspp)]
        if emissions:
            cmd += ["--emissions", str(emissions)]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        frames = [l for l in p.stdout.splitlines() if l.startswith('{"child_frame"')]
        if p.returncode != 0 or not frames:
            out.setdefault("errors", []).append("%s pass: rc %d: %s" % (tag, p.returncode, (p.stderr or p.stdout)[-300:]))
            return False
        out["frame"] = json.loads(frames[-1])
        out.setdefault("pass_wall_s", {})[tag] = time.perf_counter() - t0
        for db in glob.glob(os.path.join(tmp, "**", "*_results.db"), recursive=True):
            con = sqlite3.connect(db)
            try:
                rows = list(con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
                durs = list(con.execute("select name, total_calls, total_duration from top_kernels"))
            except sqlite3.Error as ex:
                out.setdefault("errors", []).append("%s pass: %r" % (tag, ex))
                rows, durs = [], []
            con.close()

            def bucket(kname):
                """The kernel's own name WITH its template arguments (renderKernelFlatK<768>, wfTraceKernel<PoolRays, false, 3, 3>, ...):
                the per-leg profile names the instance that ran, not a family."""
                if not any(k in kname for k in INTEGRATOR_KERNELS):
                    return None  # memsets, torch kernels, the photon pass
                return kernel_instance_name(kname)
            for kname, cname, val, n in rows:
                kk = bucket(kname)
                if kk is None:
                    continue
                got = True
                d = out["per_kernel"].setdefault(kk, {})
                d[cname] = d.get(cname, 0.0) + float(val)
                d["dispatches"] = max(d.get("dispatches", 0), int(n))
                out["counters"][cname] = out["counters"].get(cname, 0.0) + float(val)
            for kname, calls, total in durs:
                kk = bucket(kname)
                if kk is not None:
                    d = out["per_kernel"].setdefault(kk, {})
                    d["duration_ms_" + tag] = d.get("duration_ms_" + tag, 0.0) + float(total) / 1e3  # top_kernels durations are in microseconds
    except Exception as ex:
        out.setdefault("errors", []).append("%s pass: %r" % (tag, ex))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got


def collect_counters(workload, sqrtspp=None, emissions=None, passes=None, timeout=420):
    """Returns {"counters": {name: sum over the integrator dispatches of ONE frame}, "per_kernel": ..., "frame": ...}
    or {"error": ...}. Separate passes as MI355X_MICROARCH.md prescribes (the read and the write counters do not fit together)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    out = {"counters": {}, "per_kernel": {}, "frame": None}
    plan = passes or PMC_PASSES
    ok = [_pmc_pass(rocprof, tag, names, workload, sqrtspp, emissions, timeout, out) for tag, names in plan]
    if passes is None and not ok[0]:
        # the combined read + SQ pass was refused: the three separate passes of round 2
        out["counters"], out["per_kernel"] = {}, {}
        for tag, names in PMC_FALLBACK:
            _pmc_pass(rocprof, tag, names, workload, sqrtspp, emissions, timeout, out)
    return out


def read_bytes_of(c):
    """Bytes the L2s asked the fabric for (HBM or Infinity Cache), from whichever read counters the pass had."""
    if "TCC_EA0_RDREQ_sum" in c:
        n, n32, n128 = c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        return 32.0 * n32 + 128.0 * n128 + 64.0 * (n - n32 - n128)
    if "FETCH_SIZE" in c:
        return c["FETCH_SIZE"] * 1024.0 * 2.0  # = the size-class sum on gfx950 (profiles/r03_traffic_calibration.json)
    return None


def counters_summary(pmc, scale=1.0):
    """Derived figures from the raw sums (scale: frame of the timed configuration / frame the counters were taken on)."""
    if not pmc or not pmc.get("counters"):
        return None
    c = pmc["counters"]
    s = {"source": "rocprofv3 --pmc passes of one frame, child processes of this run", "frame_scale": scale,
         "calibration": "profiles/r03_traffic_calibration.json: read bytes = 128-byte lines moved (= 2 x FETCH_SIZE on gfx950, exact for streams, 8 KB runs, "
                        "scattered 64-byte blocks and 80-byte records); WRITE_SIZE exact for streamed stores, 32-byte sectors for scattered ones"}
    rb = read_bytes_of(c)
    if rb is not None and "WRITE_SIZE" in c:
        s["fetch_bytes"] = rb * scale
        s["write_bytes"] = c["WRITE_SIZE"] * 1024.0 * scale
        s["traffic_bytes"] = s["fetch_bytes"] + s["write_bytes"]
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) > 0:
        s["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_ACTIVE_INST_VALU" in c and pmc.get("frame"):
        # SQ counters tick in quad-cycles (MI355X_MICROARCH.md, per-instruction constants). SIMD time is priced at the 2.4 GHz
        # spec clock over the counted frame's own kernel time, so "busy" is a share of what the chip could issue at full clock.
        frame_s = pmc["frame"]["kernel_ms"] * 1e-3
        simd_cycles = VALU_SIMDS * frame_s * 2.4e9
        s["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        s["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0) if c.get("SQ_THREAD_CYCLES_VALU") else None
        s["valu_lane_ops"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) * scale                # lane x quad-cycle = one FP64-rate lane slot
        s["valu_insts"] = c.get("SQ_INSTS_VALU", 0.0) * scale
        if c.get("SQ_WAVE_CYCLES"):
            s["waves_waiting"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        # instruction mix by operand type (wave instructions; counts, so they scale with the frame). FP64 = add/sub + mul + fma + transcendental
        # (rcp, rsq, sqrt: the divisions and square roots of the path); everything the typed counters do not name - moves, selects,
        # compares, conversions, min / max, bit operations - is "other".
        if all(k in c for k in SQ_TYPED) and c.get("SQ_INSTS_VALU"):
            f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
            f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            integer = c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"]
            total = c["SQ_INSTS_VALU"]
            s["fp64_insts"] = f64 * scale                                                 # wave instructions
            s["fp64_flop_insts"] = (f64 + c["SQ_INSTS_VALU_FMA_F64"]) * scale             # an fma is two flops
            s["inst_mix"] = {"f64": f64 / total, "f64_fma_share": c["SQ_INSTS_VALU_FMA_F64"] / max(f64, 1.0), "f32": f32 / total, "int": integer / total,
                             "other": max(0.0, 1.0 - (f64 + f32 + integer) / total)}
        if c.get("GRBM_GUI_ACTIVE"):
            clk = c["GRBM_GUI_ACTIVE"] / frame_s
            if clk > 3.0e9:  # summed over the 8 XCDs
                clk /= 8.0
            s["measured_clock_GHz"] = clk / 1e9
    if pmc.get("errors"):
        s["errors"] = pmc["errors"]
    if pmc.get("frame"):
        s["frame"] = pmc["frame"]
    if pmc.get("pass_wall_s"):
        s["pass_wall_s"] = pmc["pass_wall_s"]
    return s


def write_leg_profile(name, desc, pmc_raw, summary, roofline):
    """The counters behind a leg's roofline block as a file (gpurun_out/bench_profiles/pmc_<leg>.md; committed as
    profiles/rNN_pmc_<leg>.md): raw sums per kernel, the derived figures and the arithmetic that leads to frac / traffic."""
    try:
        os.makedirs(PROFILE_DIR, exist_ok=True)
        with open(os.path.join(PROFILE_DIR, "pmc_%s.md" % name), "w") as f:
            f.write("# rocprofv3 PMC summary of bench.py leg `%s`\n\n%s\n\n" % (name, desc))
            fr = (pmc_raw or {}).get("frame") or {}
            f.write("Counted frame (child process of bench.py under `rocprofv3 --kernel-trace --pmc ...`, one pass per counter set): "
                    "%s spp, %s rays, %s paths, %s kNN searches, kernel time %.3f ms (HIP events)\n\n"
                    % (fr.get("spp"), fr.get("rays"), fr.get("paths"), fr.get("knn_searches"), fr.get("kernel_ms", float("nan"))))
            f.write("## raw counter sums per kernel (all dispatches of the counted frame)\n\n| kernel | counter | sum |\n|---|---|---:|\n")
            for k, d in sorted(((pmc_raw or {}).get("per_kernel") or {}).items()):
                for cn, v in sorted(d.items()):
                    f.write("| `%s` | %s | %.6g |\n" % (k, cn, v))
            f.write("\n## derived (frame_scale = timed frame / counted frame)\n\n```json\n%s\n```\n" % json.dumps(summary, indent=1))
            f.write("\n## roofline block of the bench line\n\n```json\n%s\n```\n" % json.dumps(roofline, indent=1))
            f.write("\nRecompute: read bytes = 32 n32 + 128 n128 + 64 (RDREQ - n32 - n128); write bytes = WRITE_SIZE x 1024; traffic = (read + write) x frame_scale "
                    "(all kernels between the frame's HIP events: integrator kernels + sampleResolveKernel); hbm_frac_measured = traffic / kernel_ms of the timed "
                    "frame / 8 TB/s; valu_issue_frac = SQ_THREAD_CYCLES_VALU x frame_scale / kernel_ms / (1024 SIMDs x 16 lanes x 2.4 GHz); valu_busy = "
                    "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x counted kernel time x 2.4 GHz); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                    "fp64_frac = (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + FMA_F64 + TRANS_F64) x frame_scale x 64 x lane_utilisation / kernel_ms / 39.3 T lane-op/s (the FP64 peak "
                    "without fma contraction; the library is built -ffp-contract=off); inst_mix = the typed counters / SQ_INSTS_VALU; "
                    "frac = max(hbm_frac_measured, valu_issue_frac). valu_busy, waves_waiting and measured_clock_GHz are time ratios of the COUNTED frame.\n")
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    pass


def setup_workload(name, args, m, tiling, rank, world, local_rank, dist, sqrtspp=None):
    """Scene (and photon maps) resident in HBM, output tile allocated: everything the timed region needs."""
    import torch

    wl = Workload()
    wl.name = name
    image_file, W, H, s, desc = WORKLOADS[name]
    if sqrtspp:
        s = int(sqrtspp)
        desc += " [spp overridden to %d]" % (s * s)
    wl.W, wl.H, wl.sqrtspp, wl.desc = W, H, s, desc
    if name in ("c3", "c4", "c5"):
        sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
        import make_large
        if local_rank == 0 and make_large.ensure_image(name) is None:
            raise RuntimeError("%s needs oracle/_ref (python __graft_entry__.py build in the build container)" % name)
        if _collectives(world):
            dist.barrier()
    path = os.path.join(ROOT, "tests", "golden", image_file)
    if not os.path.exists(path):
        raise RuntimeError("scene image %s is not on this machine" % os.path.normpath(path))
    img = m.SceneImage(path)
    cam = img.camera
    cam.width, cam.height, cam.sqrtspp = W, H, s
    wl.img, wl.full = img, cam.copy()
    wl.cam = tiling.shard_camera(wl.full, rank, world, SHARD_ROWS)
    wl.ctx = m.Context(local_rank)
    wl.ctx.upload_image(img)  # scene resident in HBM before the timed region
    wl.integrator = m.INTEGRATOR_PATH_TRACER
    wl.pm_maps, wl.emit_info = None, None
    wl.photon = name in ("pm", "c5")
    wl.emissions = args.emissions or EMISSIONS.get(name, 1e6)
    wl.photon_allgather_ms = None
    if wl.photon and not _collectives(world) and not args.host_octree:
        # one GPU: the whole photon pass on the device (mcrt_photon_pass_device: emission, sort, octants, boxes, record lists;
        # no photon list crosses PCIe)
        wl.integrator = m.INTEGRATOR_PHOTON_MAPPER
        sc = img.scene
        t_pass = time.perf_counter()
        ps = wl.ctx.photon_pass_device(wl.emissions, 10.0, SEED, sc.bb_min[:], sc.bb_max[:], 200, 50, False)
        cold_s = time.perf_counter() - t_pass
        # once more: the pass as it costs per frame of an animation (work buffers pooled in the context, DevPool); same maps
        if not getattr(args, "child_frame", False):
            t_pass = time.perf_counter()
            ps = wl.ctx.photon_pass_device(wl.emissions, 10.0, SEED, sc.bb_min[:], sc.bb_max[:], 200, 50, False)
        wl.emit_info = dict(paths=ps["emission_paths"], rays=ps["rays"], kernel_ms=ps["emission_ms"], global_photons=int(ps["global_count"]),
                            caustic_photons=int(ps["caustic_count"]), octree_build_s=(ps["total_ms"] - ps["emission_ms"]) * 1e-3,
                            octree_builder="device (mcrt_photon_pass_device)", photon_pass_s=time.perf_counter() - t_pass, photon_pass_first_call_s=cold_s,
                            map_ms=dict(sort=ps["sort_ms"], octants=ps["octant_ms"], boxes_and_lists=ps["finish_ms"]),
                            octants=[int(ps["global_octants"]), int(ps["caustic_octants"])],
                            emission_Mray_per_s=ps["rays"] / max(ps["emission_ms"], 1e-9) / 1e3)
    elif wl.photon and not args.host_octree:
        # N > 1: every rank emits ITS shard of the emission paths, the lists stay in HBM (mcrt_emit_photons_device), ONE all-gather
        # per map moves them between the GPUs on device pointers (RCCL over xGMI; counts first, then the padded lists), and every
        # rank builds the same two maps from the concatenation (mcrt_upload_photons_device). No photon list touches the host.
        wl.integrator = m.INTEGRATOR_PHOTON_MAPPER
        dev = torch.device("cuda", local_rank)
        t_pass = time.perf_counter()
        em = wl.ctx.emit_photons_device(wl.emissions, 10.0, SEED, rank, world)
        wl.emit_info = dict(paths=em["paths"], rays=em["rays"], kernel_ms=em["kernel_ms"])

        class _DevList:  # a device pointer as a tensor, no copy
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n, 8), "typestr": "<f4", "data": (ptr, False), "version": 2}

        lists = []
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter()
        for key in ("global_", "caustic"):
            ptr, n = em[key]
            ph = torch.as_tensor(_DevList(ptr, n), device=dev) if n else torch.zeros((0, 8), dtype=torch.float32, device=dev)
            sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
            counts = [int(x.item()) for x in sizes]
            cap = max(max(counts), 1)
            pad = torch.zeros((cap, 8), dtype=torch.float32, device=dev)
            pad[:n] = ph
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            lists.append(torch.cat([parts[r][: counts[r]] for r in range(world)]).contiguous())
        sc = img.scene
        # the lists were produced on torch's streams (all_gather on RCCL's, cat / contiguous on the current one); the library builds the
        # maps on the context's own stream and include/mcrt.h requires device inputs to be COMPLETE when the call is made
        torch.cuda.synchronize(dev)
        wl.photon_allgather_ms = (time.perf_counter() - t_gather) * 1e3   # counts + padded lists of both maps, incl. the concatenation
        t_build = time.perf_counter()
        ps = wl.ctx.upload_photons_device(lists[0].data_ptr(), lists[0].shape[0], lists[1].data_ptr(), lists[1].shape[0], sc.bb_min[:], sc.bb_max[:],
                                          200, 50, False)
        torch.cuda.synchronize(dev)
        wl.emit_info.update(global_photons=int(lists[0].shape[0]), caustic_photons=int(lists[1].shape[0]),
                            octree_build_s=time.perf_counter() - t_build, octree_builder="device (mcrt_upload_photons_device), lists all-gathered on device pointers",
                            photon_pass_s=time.perf_counter() - t_pass, octants=[int(ps["global_octants"]), int(ps["caustic_octants"])],
                            emission_Mray_per_s=wl.emit_info["rays"] / max(wl.emit_info["kernel_ms"], 1e-9) / 1e3)
        del lists
    elif wl.photon:
        # --host-octree: the lists over the host, the recursive host builder (A/B runs)
        wl.integrator = m.INTEGRATOR_PHOTON_MAPPER
        em = wl.ctx.emit_photons(wl.emissions, 10.0, SEED, rank, world)
        wl.emit_info = dict(paths=em["paths"], rays=em["rays"], kernel_ms=em["kernel_ms"])
        lists = []
        for key in ("global_", "caustic"):
            ph = torch.from_numpy(em[key][0]).to(torch.device("cuda", local_rank))
            if _collectives(world):
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
        wl.pm_maps = (m.PhotonMap(lists[0], sc.bb_min[:], sc.bb_max[:], 200, ctx=None), m.PhotonMap(lists[1], sc.bb_min[:], sc.bb_max[:], 200, ctx=None))
        wl.emit_info.update(global_photons=int(lists[0].shape[0]), caustic_photons=int(lists[1].shape[0]),
                            octree_build_s=time.perf_counter() - t_build, octree_builder="host",
                            emission_Mray_per_s=wl.emit_info["rays"] / max(wl.emit_info["kernel_ms"], 1e-9) / 1e3)
        wl.ctx.upload_photons(wl.pm_maps[0].desc, wl.pm_maps[1].desc, 50, False)
    wl.my_rows = m.shard_rows(wl.cam)
    wl.dev = torch.device("cuda", local_rank)
    wl.tile = torch.zeros((tiling.max_rows(wl.full, world, SHARD_ROWS), W, 3), dtype=torch.float64, device=wl.dev)  # packed owned rows (+ padding)
    wl.gathered = [torch.empty_like(wl.tile) for _ in range(world)] if (_collectives(world) and rank == 0) else None
    wl.stream = torch.cuda.current_stream(wl.dev).cuda_stream
    return wl


def run_steps(wl, steps, warmup, world, dist, warm_sqrtspp=None):
    """W untimed steps, then exactly K timed steps between barrier + synchronize; returns (seconds, per-step stats).
    warm_sqrtspp (secondary legs whose frame takes tens of seconds): the untimed frame runs at that spp."""
    import torch

    def step(cam=None):
        wl.ctx.render_device(cam or wl.cam, SEED, wl.integrator, wl.tile.data_ptr(), wl.stream)
        st = wl.ctx.render_finish()
        if _collectives(world):
            dist.gather(wl.tile, wl.gathered, dst=0)  # the single collective of the data path
        return st

    def sync():
        if _collectives(world):
            dist.barrier()
        torch.cuda.synchronize(wl.dev)

    warm_cam = None
    if warm_sqrtspp:
        warm_cam = wl.cam.copy()
        warm_cam.sqrtspp = int(warm_sqrtspp)
    for _ in range(warmup):
        step(warm_cam)
    sync()
    t0 = time.perf_counter()
    stats = [step() for _ in range(steps)]
    sync()
    elapsed = time.perf_counter() - t0
    wl.gather_ms = None
    if _collectives(world):  # after the timed region: what the frame's one collective costs by itself (the same tiles again)
        t1 = time.perf_counter()
        for _ in range(3):
            dist.gather(wl.tile, wl.gathered, dst=0)
        sync()
        wl.gather_ms = (time.perf_counter() - t1) * 1e3 / 3.0
    return elapsed, stats


# Necessary work of the REFERENCE's algorithm on the same rays, in FP64 lane-ops (one +, -, x, /, sqrt, min, max or compare = 1): the
# oracle's per-ray test counts (best-first traversal of the reference's tree) x the operation count of each test as the reference
# writes it - BoundingBox::intersect 24 (bounding-box.cpp:9-17), Triangle::intersect 28 at the u exit ... 54 accepted, priced 41
# (triangle.cpp:23-63), Sphere::intersect 21 at the discriminant exit ... 31 accepted, priced 26 (sphere.cpp:13-26) - plus shading:
# ~640 per path vertex (Interaction, emissive, light sample + BSDF value/pdf, BSDF sample, roulette, next ray; static FP64 count of
# the shading code's common path) / 1.6 rays per vertex = 400 per ray. Photon-mapped legs add per kNN search (linear-octree.cpp:25-117,
# photon-mapper.cpp:343-391): 20 per octant visited (BoundingBox::distance2 / max_distance2, bounding-box.cpp:43-54), 9 per photon
# scanned (distance2 + compare) and 100 per photon of the estimate (Photon::dir's sincosf pair, Interaction::BSDF, the weighted sum).
NECESSARY_OPS = {"box_test": 24.0, "triangle_test": 41.0, "sphere_test": 26.0, "shading_per_ray": 400.0,
                 "knn_octant_test": 20.0, "knn_photon_test": 9.0, "knn_estimate_per_photon": 100.0, "knn_k": 50.0}


def add_physical_fractions(r, counts, kernel_ms, rays_per_launch, knn_per_launch, pmc):
    """Every leg's roofline block carries the same three fractions, each a share of a PHYSICAL peak of the chip:
      hbm_frac_measured  bytes the L2s moved over the fabric (rocprofv3 counters of one frame, made in this run) / kernel time / 8 TB/s
      valu_issue_frac    lane-slots that executed a vector instruction (SQ_THREAD_CYCLES_VALU) / kernel time / 39.3 T lane-op/s
                         (= VALU busy x lane utilisation; integer, move and FP32 cull instructions count: occupancy, not useful work)
      frac_necessary     FP64 lane-ops the REFERENCE's algorithm needs for the same rays (NECESSARY_OPS) / kernel time / 39.3 T lane-op/s
    `frac` = the larger of the two measured ones and `bound` / `achieved` / `peak` / `unit` name it. The algorithmic-bytes rate
     stays beside them as `algorithmic_GBs` / `algorithmic_frac`: a work rate in the reference's units - caches and
    LDS serve most of those bytes, so it can exceed 1 and is never `frac`."""
    sec = kernel_ms * 1e-3
    o = NECESSARY_OPS
    per_ray = counts["node_per_ray"] * o["box_test"] + counts["tri_per_ray"] * o["triangle_test"] + counts["sphere_per_ray"] * o["sphere_test"] + o["shading_per_ray"]
    necessary = per_ray * rays_per_launch
    r["necessary_lane_ops_per_ray"] = per_ray
    if counts.get("knn_photons_per_search") and knn_per_launch:
        per_search = counts["knn_octants_per_search"] * o["knn_octant_test"] + counts["knn_photons_per_search"] * o["knn_photon_test"] + o["knn_k"] * o["knn_estimate_per_photon"]
        r["necessary_lane_ops_per_search"] = per_search
        necessary += per_search * knn_per_launch
    r["necessary_lane_ops_prices"] = o
    r["frac_necessary"] = necessary / sec / 1e9 / VALU_PEAK_GLANEOPS
    r["hbm_frac_measured"] = r["valu_issue_frac"] = None
    if pmc and pmc.get("traffic_bytes"):
        r["traffic_GBs"] = pmc["traffic_bytes"] / sec / 1e9
        r["hbm_frac_measured"] = r["traffic_GBs"] / HBM_PEAK_GBS
        r["traffic_over_algorithmic"] = pmc["traffic_bytes"] / (r["algorithmic_GBs"] * 1e9 * sec)
    if pmc and pmc.get("valu_lane_ops"):
        r["valu_G_lane_ops_per_s"] = pmc["valu_lane_ops"] / sec / 1e9
        r["valu_issue_frac"] = r["valu_G_lane_ops_per_s"] / VALU_PEAK_GLANEOPS
    r["fp64_frac"] = None
    if pmc and pmc.get("fp64_insts") and pmc.get("lane_utilisation"):
        # the FP64 roofline of a kernel compiled -ffp-contract=off: one FP64 lane-op per lane slot, 39.3 T/s (an fma, two flops, also takes
        # one slot: the 78.6 TFLOP/s peak is out of reach without contraction). Lanes: the wave instructions counted by type x 64 x the
        # measured share of lanes that were on (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU, the kernels' average over ALL their VALU
        # instructions - the typed counters have no per-lane form)
        r["fp64_G_lane_ops_per_s"] = pmc["fp64_insts"] * 64.0 * pmc["lane_utilisation"] / sec / 1e9
        r["fp64_frac"] = r["fp64_G_lane_ops_per_s"] / VALU_PEAK_GLANEOPS
        r["fp64_TFLOPs"] = pmc["fp64_flop_insts"] * 64.0 * pmc["lane_utilisation"] / sec / 1e12
        r["fp64_peak_TFLOPs"] = VALU_PEAK_GLANEOPS / 1e3
        r["inst_mix"] = pmc.get("inst_mix")
    if pmc:
        # counts (bytes, instructions) scale with the frame and are priced over the TIMED frame; time ratios (valu_busy, waves_waiting)
        # and the clock belong to the COUNTED frame - a child process under rocprofv3, possibly at fewer samples per pixel - and are
        # reported as that frame's, never carried over (counted_frame: its spp and kernel time)
        keep = ("valu_busy", "lane_utilisation", "waves_waiting", "valu_insts", "measured_clock_GHz", "fetch_bytes", "write_bytes", "l2_hit_rate", "frame_scale",
                "pass_wall_s", "errors")
        r["counters"] = {k: pmc[k] for k in keep if k in pmc}
        if pmc.get("frame"):
            r["counters"]["counted_frame"] = {"spp": pmc["frame"].get("spp"), "kernel_ms": pmc["frame"].get("kernel_ms"),
                                              "time_ratios_are_of_this_frame": ["valu_busy", "waves_waiting", "measured_clock_GHz"]}
        if pmc.get("valu_lane_ops"):
            r["counters"]["lane_ops_per_ray"] = pmc["valu_lane_ops"] / rays_per_launch
            r["counters"]["valu_insts_per_ray"] = pmc["valu_insts"] * 64.0 / rays_per_launch  # wave instructions x 64 lanes
    hbm, valu = r["hbm_frac_measured"], r["valu_issue_frac"]
    if hbm is None and valu is None:
        r["bound"], r["frac"], r["achieved"], r["peak"], r["unit"] = "hbm", None, None, HBM_PEAK_GBS, "GB/s"
        r["note"] += "; counters unavailable in this run (%s): no measured fraction" % ((pmc or {}).get("errors") or "rocprofv3 passes skipped")
    elif valu is None or (hbm is not None and hbm >= valu):
        r["bound"], r["frac"], r["achieved"], r["peak"], r["unit"] = "hbm", hbm, r["traffic_GBs"], HBM_PEAK_GBS, "GB/s"
    else:
        r["bound"], r["frac"], r["achieved"], r["peak"] = "valu", valu, r["valu_G_lane_ops_per_s"], VALU_PEAK_GLANEOPS
        r["unit"] = "G lane-op/s (FP64 rate: 1024 SIMDs x 16 lanes x 2.4 GHz)"
    return r


def leg_roofline(m, counts, kernel_id, kernel_ms, rays_per_launch, knn_per_launch, pmc):
    """The roofline block of one leg (headline included). Algorithmic bytes: B_ray = n_node*64 + n_tri*72 +
    n_sphere*32 + 300 with the per-ray counts of the reference-equivalent best-first traversal (oracle, rows of this frame) and
    B_knn = n_octant*128 + n_photon_scanned*32 + k*32 per search; the measured fractions: add_physical_fractions."""
    b_ray = counts["node_per_ray"] * 64 + counts["tri_per_ray"] * 72 + counts["sphere_per_ray"] * 32 + 300
    sec = kernel_ms * 1e-3
    alg = rays_per_launch * b_ray / sec / 1e9
    r = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
         "traffic": pmc.get("traffic_bytes") if pmc else None,
         "kernel": m.KERNEL_NAMES.get(kernel_id, "?") + (" (all launches of the frame)" if kernel_id in (m.KERNEL_WAVEFRONT, m.KERNEL_WAVEFRONT_PM) else ""),
         "kernel_id": kernel_id, "kernel_ms": kernel_ms, "bytes_per_ray": b_ray, "rays_per_launch": rays_per_launch,
         "per_ray_counts": {k: counts[k] for k in ("node_per_ray", "tri_per_ray", "sphere_per_ray")},
         "note": "frac = the larger of hbm_frac_measured and valu_issue_frac (both from rocprofv3 PMC passes of one frame made in this run, priced over the "
                 "HIP-event kernel time of the timed steps); algorithmic_GBs = reference-equivalent bytes (SURVEY.md 8(d)) per second of kernel time: a work "
                 "rate, served mostly by LDS / L2 / Infinity Cache, reported beside the bound and never as its fraction; traffic = measured fabric bytes of one frame"}
    if counts.get("knn_photons_per_search") and knn_per_launch:
        b_knn = counts["knn_octants_per_search"] * 128 + counts["knn_photons_per_search"] * 32 + 50 * 32
        knn_rate = knn_per_launch / sec
        r["knn"] = {"bytes_per_search": b_knn, "searches_per_s_in_kernel": knn_rate}
        r["algorithmic_GBs_rays_only"] = alg
        alg += knn_rate * b_knn / 1e9
    r["algorithmic_GBs"] = alg
    r["algorithmic_frac"] = alg / HBM_PEAK_GBS
    return add_physical_fractions(r, counts, kernel_ms, rays_per_launch, knn_per_launch, pmc)



# ------------------------------------------------------------------------------------------------------------------
# the line the driver parses: short (the driver's record keeps ~8 KB of stdout), the LAST line of stdout
# ------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 8000           # characters; tests/test_bench_line.py holds compact_line() to it
FULL_RECORD = os.path.join(PROFILE_DIR, "bench_full.json")


def _rnd(v, sig=6):
    """Floats to `sig` significant digits (the full-precision record is bench_full.json)."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _rnd(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_rnd(x, sig) for x in v]
    return v


ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_ray", "rays_per_launch",
                 "algorithmic_GBs", "algorithmic_frac", "hbm_frac_measured", "traffic_GBs", "valu_issue_frac", "fp64_frac", "fp64_TFLOPs",
                 "frac_necessary", "traffic_over_algorithmic")
COUNTER_KEYS = ("valu_busy", "lane_utilisation", "waves_waiting", "measured_clock_GHz", "l2_hit_rate", "frame_scale", "lane_ops_per_ray")  # (of the counted frame)
LEG_ROOFLINE_KEYS = ("bound", "frac", "hbm_frac_measured", "valu_issue_frac", "fp64_frac", "algorithmic_frac", "frac_necessary", "kernel_ms")


def compact_roofline(r, keys=ROOFLINE_KEYS, counters=COUNTER_KEYS):
    if not r:
        return None
    out = {k: r[k] for k in keys if r.get(k) is not None or k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if "unit" in out and out["unit"] and out["unit"].startswith("G lane-op/s"):
        out["unit"] = "G lane-op/s"     # FP64-rate lane slots: 1024 SIMDs x 16 lanes x 2.4 GHz
    c = r.get("counters") or {}
    for k in counters:
        if c.get(k) is not None:
            out[k] = c[k]
    if r.get("inst_mix"):
        out["inst_mix"] = {k: round(v, 4) for k, v in r["inst_mix"].items()}
    if (c.get("counted_frame") or {}).get("spp") is not None:
        out["counted_spp"] = c["counted_frame"]["spp"]
    if c.get("errors"):
        out["counter_errors"] = str(c["errors"])[:200]
    return out


def compact_cpu(b, sample_chars=150):
    if not b:
        return None
    out = {k: b[k] for k in ("value", "unit", "cores", "kind") if k in b}
    out["sample"] = (b.get("sample") or "")[:sample_chars]
    for k in ("best_value", "best_cores", "port_value", "port_cores", "knn_searches_per_s", "quoted"):
        if b.get(k) is not None:
            out[k] = b[k]
    return out


def compact_line(result):
    """The headline line: the contract's keys, `config`, `parity`, `cpu_baseline`, `roofline` of the headline leg, and per secondary leg a
    dozen numbers. Everything else (per-leg counters, prices, notes, thread scans, photon-pass breakdowns) is in bench_full.json /
    profiles/. Stays under LINE_LIMIT characters whatever the legs hold: optional keys are shed in a fixed order if it does not."""
    core = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: result.get(k) for k in core}
    cfg = result.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "width", "height", "spp", "integrator", "seed", "sharding", "rays_per_step", "paths_per_step",
                                             "kernel", "kernel_launches_per_step", "frame_finite", "frame_mean_radiance") if k in cfg}
    for k in ("per_rank", "gather_ms", "photon_allgather_ms", "frame_with_photon_pass_ms"):
        if result.get(k) is not None:
            out[k] = result[k]
    if result.get("parity"):
        out["parity"] = {k: result["parity"].get(k) for k in ("rows", "pixels", "max_rel", "outliers_gt_1e-4", "bit_identical", "tolerance")}
    out["cpu_baseline"] = compact_cpu(result.get("cpu_baseline"))
    out["roofline"] = compact_roofline(result.get("roofline"))
    legs = {}
    for name, leg in (result.get("secondary") or {}).items():
        if "error" in leg:
            legs[name] = {"error": str(leg["error"])[:160]}
            continue
        c, r, b = leg.get("config") or {}, leg.get("roofline") or {}, leg.get("cpu_baseline") or {}
        e = {"workload": (c.get("workload") or "")[:60], "value": leg.get("value"), "unit": leg.get("unit"), "steps": leg.get("steps"), "ms_per_step": leg.get("ms_per_step"),
             "kernel": c.get("kernel"), "launches_per_step": c.get("kernel_launches_per_step")}
        for k in ("frame_with_photon_pass_ms", "value_with_photon_pass"):
            if leg.get(k) is not None:
                e[k] = leg[k]
        if c.get("knn_searches_per_s"):
            e["knn_searches_per_s"] = c["knn_searches_per_s"]
        if leg.get("parity"):
            e["bit_identical"] = leg["parity"].get("bit_identical")
        e.update({k: r[k] for k in LEG_ROOFLINE_KEYS if r.get(k) is not None})
        for k in ("lane_utilisation", "measured_clock_GHz"):
            if (r.get("counters") or {}).get(k) is not None:
                e[k] = r["counters"][k]
        if b:
            e["cpu"] = {"value": b.get("value"), "cores": b.get("cores"), "kind": b.get("kind")}
        legs[name] = e
    if legs:
        out["secondary"] = legs
    tol = result.get("tolerance_build")
    if tol:
        out["tolerance_build"] = {}
        for name, e in tol.items() if isinstance(tol, dict) and "error" not in tol else []:
            if "error" in e:
                out["tolerance_build"][name] = {"error": str(e["error"])[:120]}
                continue
            t = {"value": e.get("value"), "ms_per_step": e.get("ms_per_step"), "vs_exact_build": e.get("vs_exact_build")}
            if e.get("parity"):
                t.update({k: e["parity"].get(k) for k in ("max_rel", "p999_rel", "outliers_gt_1e-4")})
            out["tolerance_build"][name] = t
        if isinstance(tol, dict) and "error" in tol:
            out["tolerance_build"] = {"error": str(tol["error"])[:120]}
    out["detail"] = "gpurun_out/bench_profiles/bench_full.json (+ pmc_<leg>.md); committed as profiles/rNN_bench_full.json"
    out = _rnd(out)
    # shed optional keys, least important first, until the line fits (never the contract's keys)
    shed = [("secondary", k) for k in ("workload", "measured_clock_GHz", "lane_utilisation", "frac_necessary", "launches_per_step", "cpu", "kernel_ms")] + \
           [("roofline", k) for k in ("traffic_over_algorithmic", "lane_ops_per_ray", "l2_hit_rate", "waves_waiting", "frac_necessary")]
    line = json.dumps(out)
    while len(line) > LINE_LIMIT and shed:
        where, key = shed.pop(0)
        if where == "secondary":
            for e in out.get("secondary", {}).values():
                e.pop(key, None)
        elif out.get(where):
            out[where].pop(key, None)
        line = json.dumps(out)
    if len(line) > LINE_LIMIT:
        out.pop("secondary", None)
        line = json.dumps(out)
    return line

PARITY_ROWS = {"c2": "hexagon_room.c2_1920x1080_s16_rows536_540.f64", "c2_ggx": "hexagon_room_ggx.c2ggx_1920x1080_s16_rows536_540.f64"}


def c2_parity(wl, frame):
    """Rows 536-540 of the full-size C2 / C2-GGX frame against the reference's own radiance (tests/golden fixtures, made by
    tests/golden/make_golden.py with the reference itself)."""
    g = os.path.join(ROOT, "tests", "golden", PARITY_ROWS.get(wl.name, "-"))
    if wl.sqrtspp != 16 or not os.path.exists(g):
        return None
    ref = np.fromfile(g, dtype=np.float64).reshape(4, 1920, 3)
    out = frame[536:540].cpu().numpy()
    rel = (np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)).max(axis=2)
    return {"rows": [536, 540], "pixels": int(rel.size), "max_rel": float(rel.max()), "p999_rel": float(np.quantile(rel, 0.999)),
            "outliers_gt_1e-4": int((rel > 1e-4).sum()), "bit_identical": bool(np.array_equal(out, ref)), "tolerance": 1e-4,
            "reference": "tests/golden/%s (rendered by the reference)" % PARITY_ROWS[wl.name]}


def measure(name, args, m, tiling, rank, world, local_rank, dist, steps, warmup, want_cpu, want_counters, headline, ref_threads=None):
    """One leg: set up, time, describe. Returns (result dict on rank 0 | None, best reference thread count)."""
    import torch

    plan = {} if headline else LEG_PLAN.get(name, {})
    steps = plan.get("steps") or steps
    wl = setup_workload(name, args, m, tiling, rank, world, local_rank, dist, sqrtspp=args.sqrtspp if headline else None)
    elapsed, stats = run_steps(wl, steps, warmup, world, dist, warm_sqrtspp=plan.get("warm_sqrtspp"))
    dev = wl.dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    acc = torch.tensor([float(sum(s["rays"] for s in stats)), float(sum(s["paths"] for s in stats)),
                        float(sum(s["kernel_ms"] for s in stats)), float(sum(s["knn_searches"] for s in stats))], dtype=torch.float64, device=dev)
    per_rank = None
    if _collectives(world):
        mine = torch.tensor([elapsed / steps * 1e3, float(sum(s["kernel_ms"] for s in stats)) / steps, wl.gather_ms or 0.0, wl.photon_allgather_ms or 0.0],
                            dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = {"ms_per_step": [float(e[0]) for e in every], "kernel_ms": [float(e[1]) for e in every],
                    "gather_ms": max(float(e[2]) for e in every), "photon_allgather_ms": max(float(e[3]) for e in every) or None}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kmax = acc[2:3].clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc[2] = kmax[0]
    elapsed = float(t.item())
    total_rays, total_paths, kernel_ms_sum, total_knn = (float(x) for x in acc)
    result = None
    if rank == 0:
        if _collectives(world):  # sanity: the gathered frame is complete and finite
            frame = torch.zeros((wl.H, wl.W, 3), dtype=torch.float64, device=dev)
            for r in range(world):
                rows = torch.from_numpy(tiling.rows_of(wl.full, r, world, SHARD_ROWS)).to(dev)
                frame[rows] = wl.gathered[r][: len(rows)]
        else:
            frame = wl.tile[: len(wl.my_rows)]
        kernel_id = stats[-1]["kernel_id"]
        result = {
            "metric": "Mray/s (whole node), 1920x1080 @ 256 spp path trace" if name.startswith("c2") else "Mray/s (whole node)",
            "value": total_rays / elapsed / 1e6, "unit": "Mray/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "reference scene" if name in REFERENCE_SCENE_LEGS else "reference scene, synthetic stand-in meshes",
            "config": {"workload": wl.desc, "width": wl.W, "height": wl.H, "spp": wl.sqrtspp ** 2, "integrator": "photon_mapper" if wl.photon else "path_tracer",
                       "seed": SEED, "sharding": "rows in groups of %d, round-robin over %d GPU(s)" % (SHARD_ROWS, world),
                       "rays_per_step": total_rays / steps, "paths_per_step": total_paths / steps,
                       "frame_mean_radiance": float(frame.mean().item()), "frame_finite": bool(torch.isfinite(frame).all().item()),
                       "kernel": m.KERNEL_NAMES.get(kernel_id, "?") + (" [lean instance: compiled without the material branches this scene does not use]"
                                                                          if wl.ctx.get_option("MCRT_LEAN_USED") == "1" else ""),
                       "kernel_id": kernel_id, "kernel_launches_per_step": stats[-1]["kernel_launches"],
                       "knn_searches_per_s": total_knn / elapsed if total_knn else None,
                       "photon_pass": wl.emit_info if wl.photon else None},
        }
        if wl.photon and wl.emit_info and wl.emit_info.get("photon_pass_s") is not None:
            # PhotonMapper's constructor (photon-mapper.cpp:24-223) is part of what BASELINE configs[4] names: the frame with its photon
            # pass (emission + both maps, on the device) beside the eye-pass figure the step times
            pp_ms = wl.emit_info["photon_pass_s"] * 1e3
            result["frame_with_photon_pass_ms"] = result["ms_per_step"] + pp_ms
            result["value_with_photon_pass"] = (total_rays / steps + wl.emit_info["rays"]) / (result["frame_with_photon_pass_ms"] * 1e-3) / 1e6
        if per_rank:
            result["per_rank"] = {k: per_rank[k] for k in ("ms_per_step", "kernel_ms")}
            result["gather_ms"] = per_rank["gather_ms"]
            if per_rank["photon_allgather_ms"]:
                result["photon_allgather_ms"] = per_rank["photon_allgather_ms"]
            if REHEARSE_DIST:
                result["rehearsal"] = "collectives of the N > 1 path run with a process group of one rank (--rehearse-dist): not a scaling measurement"
        par = c2_parity(wl, frame)  # any world: rank 0 holds the assembled frame of the last timed step
        if par:
            result["parity"] = par
        counts = None
        if world == 1 and want_cpu:
            if wl.photon and wl.pm_maps is None:  # device-built maps: host copies for the CPU leg only (same maps the GPU searched)
                wl.pm_maps = (wl.ctx.download_map(0), wl.ctx.download_map(1))
            base, counts = cpu_baseline(m, wl.img, wl.full, args.cpu_seconds if headline else args.cpu_seconds * 0.3, wl.integrator, wl.pm_maps,
                                        scan_threads=headline, ref_threads=ref_threads, ref_emissions=REF_EMISSIONS.get(name) if wl.photon else None)
            result["cpu_baseline"] = base
            ref_threads = base.get("best_cores", ref_threads)
        counts_source = "oracle, rows of this frame, this run"
        if counts is None:
            counts = STORED_COUNTS.get(name, DEFAULT_COUNTS)
            counts_source = "stored (cpu leg skipped: --no-cpu or N > 1): oracle counts of an earlier run of the same frame"
    # the context's memory goes back before the counter passes (child processes) need the GPU
    frame = None
    wl.ctx.close()
    del wl.tile, wl.gathered
    torch.cuda.empty_cache()
    if rank == 0:
        pmc = None
        if world == 1 and want_counters:
            # counters of ONE frame; C3's frame takes 8 s under nothing and longer under counters: taken at 64 spp and scaled
            # (per-sample work is the same; stated in frame_scale)
            scale_spp = plan.get("pmc_sqrtspp")
            raw = collect_counters(name, sqrtspp=scale_spp or (args.sqrtspp if headline else None), emissions=wl.emissions if wl.photon else None)
            scale = (wl.sqrtspp / scale_spp) ** 2 if scale_spp else 1.0
            pmc = counters_summary(raw, scale) or {"errors": raw.get("errors") or [raw.get("error")]}
        launches = steps
        kernel_ms = kernel_ms_sum / launches
        rays_per_launch = total_rays / steps / world
        result["roofline"] = leg_roofline(m, counts, kernel_id, kernel_ms, rays_per_launch, total_knn / steps / world, pmc)
        result["roofline"]["per_ray_counts_source"] = counts_source
        if world == 1 and want_counters:
            write_leg_profile(name, wl.desc, raw, pmc, result["roofline"])
    return result, ref_threads


def rows_parity(m, name):
    """Full-size golden rows of a large leg (C3: rows 540-542 of 1080p @ 1024 spp; C5: row 500 at 256 spp on the image's own photon map)
    against the REFERENCE's radiance (tests/golden, made by integration/large_scenes/make_large.py with the reference itself)."""
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    cfg_name = {"c3": "c3", "c4": "c4", "c5": "c5_s16"}.get(name)
    if cfg_name is None or cfg_name not in make_large.CONFIGS:
        return None
    c = make_large.CONFIGS[cfg_name]
    p, g = make_large.ensure_image(cfg_name), make_large.golden_path(cfg_name)
    if p is None or not os.path.exists(g):
        return None
    img = m.SceneImage(p)
    ctx = m.Context(0)
    ctx.upload_image(img)
    cam = img.camera
    cam.sqrtspp = c["sqrtspp"]
    r0, r1 = c["rows"]
    cam.shard_rows, cam.shard_count = r1 - r0, (cam.height + r1 - r0 - 1) // (r1 - r0)
    cam.shard_index = r0 // (r1 - r0)
    integ = m.INTEGRATOR_PATH_TRACER
    if c["photon"]:
        integ = m.INTEGRATOR_PHOTON_MAPPER
        ctx.upload_photons(img.photons(0), img.photons(1), int(img.param("k_nearest_photons")), bool(img.param("direct_visualization")))
    out, _ = ctx.sample_image(cam, SEED, integ)
    ctx.close()
    ref = np.fromfile(g).reshape(r1 - r0, c["width"], 3)
    got = out[r0:r1]
    rel = (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)).max(axis=2)
    return {"rows": [r0, r1], "pixels": int(rel.size), "max_rel": float(rel.max()), "p999_rel": float(np.quantile(rel, 0.999)),
            "outliers_gt_1e-4": int((rel > 1e-4).sum()), "bit_identical": bool(np.array_equal(got, ref)), "tolerance": 1e-4,
            "reference": "tests/golden/%s (rendered by the reference)" % os.path.basename(g)}


def child_leg(args):
    """One leg timed in a process of its own and printed as one JSON line: how the parent measures the TOLERANCE build (the parent has
    the exact library loaded; MCRT_TOLERANCE_BUILD=1 in this process's environment selected libmcrt_hip_tol.so at import)."""
    import torch

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    torch.cuda.set_device(0)
    if args.workload in TOLERANCE_PLAN:  # short legs: the run as a whole has to stay within minutes
        LEG_PLAN[args.workload] = dict(LEG_PLAN.get(args.workload, {}), **TOLERANCE_PLAN[args.workload])
    leg, _ = measure(args.workload, args, m, tiling, 0, 1, 0, None, args.steps, args.warmup, want_cpu=False, want_counters=False,
                     headline=args.workload == "c2")
    if "parity" not in leg:
        par = rows_parity(m, args.workload)
        if par:
            leg["parity"] = par
    leg["library"] = os.path.basename(m.LIB_PATH)
    print(json.dumps({"child_leg": args.workload, "leg": leg}), flush=True)


TOLERANCE_LEGS = ("c2", "c3", "c5")   # the opt-in build's figures beside the exact build's: headline, pipeline, photon mapper
TOLERANCE_PLAN = {"c3": dict(steps=1, warm_sqrtspp=None), "c5": dict(steps=2, warm_sqrtspp=None)}   # (c2: 5 timed frames after one warm-up frame)


def tolerance_legs(args, exact):
    """The same legs through libmcrt_hip_tol.so (-ffp-contract=fast + the platform's libm: monte-carlo-ray-tracer_amd/build.py), each
    in a child process, with the parity block against the reference's own rows at BASELINE.json's 1e-4 bar. `exact` = the exact
    build's legs of this run, for the ratio. Reported, never the headline."""
    out = {}
    lib_tol = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "libmcrt_hip_tol.so")
    if not os.path.exists(lib_tol):
        return {"error": "libmcrt_hip_tol.so not built"}
    for name in TOLERANCE_LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--child-leg", "--workload", name, "--steps", str(min(args.steps, 5) if name == "c2" else args.secondary_steps),
               "--warmup", "1"]
        if args.emissions:
            cmd += ["--emissions", str(args.emissions)]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MCRT_TOLERANCE_BUILD="1"))
            lines = [l for l in p.stdout.splitlines() if l.startswith('{"child_leg"')]
            if p.returncode != 0 or not lines:
                out[name] = {"error": "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
                continue
            leg = json.loads(lines[-1])["leg"]
            e = {"value": leg["value"], "unit": leg["unit"], "steps": leg["steps"], "ms_per_step": leg["ms_per_step"], "library": leg.get("library"),
                 "build": "-ffp-contract=fast -DMCRT_PLATFORM_LIBM (MCRT_TOLERANCE_BUILD=1)"}
            if leg.get("frame_with_photon_pass_ms") is not None:
                e["frame_with_photon_pass_ms"] = leg["frame_with_photon_pass_ms"]
            if leg.get("parity"):
                e["parity"] = {k: leg["parity"].get(k) for k in ("rows", "pixels", "max_rel", "p999_rel", "outliers_gt_1e-4", "bit_identical", "tolerance")}
            ex = exact.get(name)
            if ex and ex.get("value"):
                e["vs_exact_build"] = leg["value"] / ex["value"]
            out[name] = e
        except Exception as ex:
            out[name] = {"error": repr(ex)}
    return out


def child_frame(args):
    """One frame of a workload and nothing else: the process rocprofv3 wraps for the counter passes."""
    import torch

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    torch.cuda.set_device(0)
    wl = setup_workload(args.workload, args, m, tiling, 0, 1, 0, None, sqrtspp=args.sqrtspp)
    _, stats = run_steps(wl, 1, 0, 1, None)
    st = stats[0]
    print(json.dumps({"child_frame": args.workload, "spp": wl.sqrtspp ** 2, "rays": st["rays"], "paths": st["paths"], "kernel_ms": st["kernel_ms"],
                      "kernel_id": st["kernel_id"], "knn_searches": st["knn_searches"]}), flush=True)
    wl.ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs (stored per-ray counts are used for the algorithmic bytes)")
    ap.add_argument("--no-secondary", action="store_true", help="headline only")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 PMC child passes")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sqrtspp", type=int, default=None, help="override the workload's samples per pixel (debugging; the line says so)")
    ap.add_argument("--host-octree", action="store_true", help="build the photon octrees with the host builder instead of the GPU-assisted one")
    ap.add_argument("--emissions", type=float, default=None,
                    help="photon_map.emissions of the photon-mapped workloads (x caustic_factor 10 paths); default: pm 1e6, c5 1e7 (BASELINE configs[4]: 1e8 emission paths)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="one GPU: run the collectives of the N > 1 path (RCCL init, gathers, reductions, destroy) with a process group of one rank")
    ap.add_argument("--child-frame", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-tolerance", action="store_true", help="skip the legs of the opt-in tolerance build (libmcrt_hip_tol.so)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.child_frame:
        return child_frame(args)
    if args.child_leg:
        return child_leg(args)
    # From here on file descriptor 1 is the process's stderr: whatever a library prints to "stdout" - RCCL's version banner sits in C
    # stdio's buffer until the process exits, AFTER anything this script printed (that cost round 5's first rehearsal its last line) -
    # cannot follow, precede or interleave with the one line the driver parses. That line goes to the real stdout, kept here.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # MCRT_BENCH_SHARE_GPU=1 (rehearsal of the N > 1 path on a one-GPU box, not a measurement): every rank uses cuda:0 and the
    # collectives run over gloo, which moves CUDA tensors through the host
    share_gpu = world > 1 and os.environ.get("MCRT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    global REHEARSE_DIST
    REHEARSE_DIST = bool(args.rehearse_dist) and world == 1
    if REHEARSE_DIST:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if _collectives(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    single = world == 1 and not REHEARSE_DIST   # (the rehearsal is the N > 1 line's shape: no CPU leg, no counter passes, no secondary legs)
    result, ref_threads = measure(args.workload, args, m, tiling, rank, world, local_rank, dist, args.steps, args.warmup,
                                  want_cpu=single and not args.no_cpu, want_counters=single and not args.no_counters, headline=True)
    if single and args.workload == "c2" and not args.no_secondary:
        # driver-timed legs on the kernels that walk trees in HBM / search photon maps (their own step counts: C3 is 8 s a frame)
        result["secondary"] = {}
        for name in SECONDARY:
            t0 = time.perf_counter()
            try:
                leg, _ = measure(name, args, m, tiling, rank, world, local_rank, dist, args.secondary_steps, 1,
                                 want_cpu=not args.no_cpu, want_counters=not args.no_counters, headline=False, ref_threads=ref_threads)
                leg["leg_wall_s"] = time.perf_counter() - t0
                for k in ("metric", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "n_gpus"):
                    leg.pop(k, None)
            except Exception as ex:  # a leg that cannot run here (scene image absent) must not cost the headline
                leg = {"error": repr(ex)}
            result["secondary"][name] = leg
    if single and args.workload == "c2" and not args.no_secondary and not args.no_tolerance:
        exact = dict(result.get("secondary") or {}, c2=result)
        result["tolerance_build"] = tolerance_legs(args, exact)
    # the process group goes first: whatever RCCL / torch print while it is torn down must not follow the line the driver parses
    if _collectives(world):
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the full record: a file (and stderr), never stdout - the driver parses the LAST stdout line and keeps ~8 KB of it
        try:
            os.makedirs(PROFILE_DIR, exist_ok=True)
            with open(FULL_RECORD, "w") as f:
                json.dump(result, f)
        except OSError:
            pass
        sys.stderr.write("bench.py full record: %s\n" % json.dumps(result))
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(real_stdout, (compact_line(result) + "\n").encode())


if __name__ == "__main__":
    main()
