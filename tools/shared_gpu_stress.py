#!/usr/bin/env python3
"""Does a frame ever come out wrong when several contexts render on ONE GPU at the same time?

Round 3 saw wrong frames (radiance a few per cent low, different from run to run) with an EXPERIMENTAL hit-record layout when two
contexts of one process - or three processes - shared a GPU; the layout was never committed, the fault never understood, and since
then frames of contexts that share a device are ordered (DeviceOrder, csrc/mcrt_hip.hip). This tool looks for that fault in the
COMMITTED kernels with the ordering switched off (MCRT_DEVICE_ORDER=0 in the environment):

    python tools/shared_gpu_stress.py [--procs 3] [--contexts 2] [--frames 100] [--order 0]

P processes x C contexts (one host thread each) on device 0, dirty device memory, every context renders F frames of ITS scene -
the kernel forms alternate over the contexts: flat megakernel (hexagon_room), lane state machine (coffee_maker_qsah), wavefront
pipeline (coffee_maker_qsah with MCRT_KERNEL=wf), photon-mapping megakernel (hexagon_room_pm) - and compares every frame bit for bit
with the reference's golden radiance (path-traced scenes; --scale > 1: with the CPU oracle's frame at that size) or with the context's
own first frame (photon-mapped: 1e-10 against the golden / the oracle, bit-equal among themselves). Prints one JSON line per process and a total; exit code 1 if any frame differed."""
import argparse
import importlib
import json
import os
import subprocess
import sys
import threading

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FORMS = [("hexagon_room", None, 0), ("coffee_maker_qsah", None, 0), ("coffee_maker_qsah", "wf", 0), ("hexagon_room_pm", None, 1)]


def worker(args):
    import numpy as np
    import torch
    from conftest import camera_for, golden_path, load_radiance

    pkg = importlib.import_module("monte-carlo-ray-tracer_amd")
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    junk = [torch.full((1 << 30,), 0xFF, dtype=torch.uint8, device="cuda:0") for _ in range(2)]  # what the library allocates next is dirty
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    jobs = []
    for c in range(args.contexts):
        name, kernel, integ = FORMS[(args.index * args.contexts + c) % len(FORMS)]
        case = man["cases"][name]
        img = pkg.SceneImage(golden_path(case["image"]))
        r = case["renders"][0]
        cam = camera_for(img, r)
        ref = load_radiance(r)
        if args.scale > 1:  # larger frames (longer kernels, more overlap between the contexts): the expected frame from the CPU oracle
            import oracle_lib  # checker (tools/ and tests/ only)
            cam.width, cam.height = cam.width * args.scale, cam.height * args.scale
            ref, _ = oracle_lib.render(img, cam, man["seed"], integ, threads=max(8, oracle_lib.hardware_threads() // args.procs))
        ctx = pkg.Context(0)
        if kernel:
            ctx.set_option("MCRT_KERNEL", kernel)
        ctx.upload_image(img)
        integrator = pkg.INTEGRATOR_PHOTON_MAPPER if integ else pkg.INTEGRATOR_PATH_TRACER
        first, st = ctx.sample_image(cam, man["seed"], integrator)  # (other processes may already be rendering)
        jobs.append(dict(ctx=ctx, cam=cam, integrator=integrator, ref=ref, first=first, exact=not integ, name=name + (":" + kernel if kernel else ""),
                         kernel_id=st["kernel_id"], bad=0, bad_vs_first=0, worst=0.0))
    seed = man["seed"]

    def run(j):
        for _ in range(args.frames):
            out, _ = j["ctx"].sample_image(j["cam"], seed, j["integrator"])
            if not np.array_equal(out, j["first"]):
                j["bad_vs_first"] += 1
            if j["exact"]:
                if not np.array_equal(out, j["ref"]):
                    j["bad"] += 1
            else:
                rel = float((np.abs(out - j["ref"]) / np.maximum(np.abs(j["ref"]), 1e-3)).max())
                j["worst"] = max(j["worst"], rel)
                if rel > 1e-10:
                    j["bad"] += 1

    th = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    rec = dict(process=args.index, frames_per_context=args.frames,
               contexts=[dict(scene=j["name"], kernel_id=j["kernel_id"], frames_not_the_reference=j["bad"], frames_unlike_the_first=j["bad_vs_first"],
                              first_frame_is_the_reference=bool(np.array_equal(j["first"], j["ref"])) if j["exact"] else None, worst_rel=j["worst"]) for j in jobs])
    print(json.dumps(rec), flush=True)
    for j in jobs:
        j["ctx"].close()
    return sum(j["bad"] + j["bad_vs_first"] for j in jobs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=3)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--order", type=int, default=0, help="1: leave DeviceOrder on (the production setting)")
    ap.add_argument("--scale", type=int, default=1, help="render at this multiple of the golden frames' width and height; the expected frames then come from the CPU oracle")
    ap.add_argument("--index", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.index >= 0:
        sys.exit(1 if worker(args) else 0)
    env = dict(os.environ, MCRT_DEVICE_ORDER=str(args.order))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--procs", str(args.procs), "--contexts", str(args.contexts), "--frames", str(args.frames),
                               "--scale", str(args.scale), "--index", str(i)], env=env, stdout=subprocess.PIPE, text=True) for i in range(args.procs)]
    bad = 0
    for p in procs:
        out, _ = p.communicate()
        sys.stdout.write(out)
        bad += p.returncode != 0
    total = args.procs * args.contexts * args.frames
    print(json.dumps(dict(device_order=bool(args.order), processes=args.procs, contexts_per_process=args.contexts, concurrent_frames=total,
                          processes_with_a_wrong_frame=bad)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
