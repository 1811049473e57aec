"""The N > 1 path with the HIP library under torch.distributed: one process per rank, every rank on cuda:0 of the one-GPU
box, collectives over gloo (what bench.py does with MCRT_BENCH_SHARE_GPU=1). Each rank renders ITS rows with
mcrt_render_device into a device tile, rank 0 assembles the frame with tiling.gather_frame — the same calls, in the same
order, as bench.py's world > 1 branches; only the backend differs from the driver's 8-GPU run (gloo instead of nccl).

The assembled frame must be the single-process frame bit for bit (per-pixel seeding by absolute pixel index, camera/camera.cpp:73;
photon mapper too: the order in which an estimate's photons are found and summed depends on the query alone)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, TESTS, camera_for, golden_path, rel_error

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, image, render, seed, photon, out_path):
    for p in (ROOT, TESTS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    tiling = importlib.import_module("monte-carlo-ray-tracer_amd.tiling")
    img = m.SceneImage(image)
    cam = img.camera
    cam.width, cam.height, cam.sqrtspp = render["width"], render["height"], render["sqrtspp"]
    shard = tiling.shard_camera(cam, rank, world)
    ctx = m.Context(0)
    ctx.upload_image(img)
    mode = m.INTEGRATOR_PATH_TRACER
    if photon:
        mode = m.INTEGRATOR_PHOTON_MAPPER
        ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons") or 50, bool(img.param("direct_visualization")))
    tile = torch.zeros((tiling.max_rows(cam, world), cam.width, 3), dtype=torch.float64, device="cuda:0")
    ctx.render_device(shard, seed, mode, tile.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = ctx.render_finish()
    assert st["paths"] == len(m.shard_rows(shard)) * cam.width * cam.sqrtspp ** 2
    frame = tiling.gather_frame(tile, cam, rank, world, dist)
    counts = torch.tensor([float(st["rays"]), float(st["paths"])], dtype=torch.float64)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out_path, frame.cpu().numpy())
        with open(out_path + ".json", "w") as f:
            json.dump({"rays": counts[0].item(), "paths": counts[1].item(), "kernel_id": st["kernel_id"]}, f)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


# The last case is the shape of BASELINE configs[3] on N GPUs (bench.py --gpus N --workload c4): the wavefront pipeline, and a
# per-sample store so small that every rank goes through its rows in MANY passes (integrator launches + resolve per pass).
@pytest.mark.parametrize("name,world,photon,env", [("hexagon_room", 2, False, {}), ("hexagon_room", 3, False, {}), ("coffee_maker_qsah", 2, False, {}),
                                                   ("hexagon_room_pm", 2, True, {}), ("hexagon_room_pm", 3, True, {}),
                                                   ("coffee_maker_qsah", 3, False, {"MCRT_KERNEL": "wf", "MCRT_SAMPLE_STORE_GB": "0.0002"})])
def test_ranks_sharing_the_gpu_assemble_the_single_rank_frame(pkg, manifest, tmp_path, name, world, photon, env, monkeypatch):
    import torch.multiprocessing as mp

    for k, v in env.items():
        monkeypatch.setenv(k, v)  # the ranks inherit it (spawn), mcrt_create seeds their options from it
    case = manifest["cases"][name]
    image = golden_path(case["image"])
    r = case["renders"][0]
    out_path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), image, r, manifest["seed"], photon, out_path), nprocs=world, join=True)
    frame = np.load(out_path)
    info = json.load(open(out_path + ".json"))

    img = pkg.SceneImage(image)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    mode = pkg.INTEGRATOR_PATH_TRACER
    if photon:
        mode = pkg.INTEGRATOR_PHOTON_MAPPER
        ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons") or 50, bool(img.param("direct_visualization")))
    base, st = ctx.sample_image(camera_for(img, r), manifest["seed"], mode)
    ctx.close()
    if env:
        assert st["kernel_launches"] > 20  # several passes, each a chain of shade / trace launches
    assert info["paths"] == st["paths"] and info["rays"] == st["rays"] and info["kernel_id"] == st["kernel_id"]
    np.testing.assert_array_equal(frame, base)  # (photon-mapped frames too: an estimate's sum depends on the query alone)


def _bench(gpus, extra, env=None):
    """bench.py as the driver launches it (torch.distributed.run for N > 1), ranks sharing cuda:0 over gloo."""
    cmd = [sys.executable]
    if gpus > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + extra
    e = dict(os.environ, MCRT_BENCH_SHARE_GPU="1")
    e.update(env or {})
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    # the driver's line: short, the LAST line of stdout; the full record (what these tests compare) goes to stderr and bench_full.json
    assert p.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8000, p.stdout[-600:]
    short = json.loads(lines[0])
    full = [l for l in p.stderr.splitlines() if l.startswith("bench.py full record: ")]
    assert len(full) == 1
    full = json.loads(full[0][len("bench.py full record: "):])
    assert abs(short["value"] - full["value"]) <= 1e-5 * full["value"] and short["n_gpus"] == full["n_gpus"] and short["steps"] == full["steps"]
    full["_line"] = short
    return full


@pytest.mark.parametrize("workload,extra", [("c1", []), ("pm", ["--emissions", "20000"])])
def test_bench_n2_line_equals_n1_work(workload, extra):
    """bench.py --gpus 2 (row sharding, sharded photon emission + all-gather of the lists, the gather, frame assembly on rank
    0, max-over-ranks timing, the one JSON line) against --gpus 1 on the same workload: same paths, same rays, same frame."""
    common = ["--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu", "--no-counters", "--no-secondary"] + extra
    one = _bench(1, common)
    two = _bench(2, common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["steps"] == 2 and two["scaling"] == "strong"
    assert two["config"]["paths_per_step"] == one["config"]["paths_per_step"]
    assert two["config"]["frame_finite"] and two["value"] > 0 and two["ms_per_step"] > 0
    if workload == "pm":
        # the union of the ranks' photon lists is the unsharded list (tests/test_photon_emission.py); estimates sum in search order
        assert two["config"]["photon_pass"]["global_photons"] == one["config"]["photon_pass"]["global_photons"]
        assert abs(two["config"]["frame_mean_radiance"] - one["config"]["frame_mean_radiance"]) <= 1e-9 * abs(one["config"]["frame_mean_radiance"])
        assert abs(two["config"]["rays_per_step"] - one["config"]["rays_per_step"]) <= 1e-6 * one["config"]["rays_per_step"]
    else:
        assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]
        assert two["config"]["frame_mean_radiance"] == one["config"]["frame_mean_radiance"]
    assert "roofline" in two and two["roofline"]["kernel_id"] == one["roofline"]["kernel_id"]


@pytest.mark.parametrize("workload,extra", [("c1", []), ("pm", ["--emissions", "20000"])])
def test_bench_rccl_branch_rehearsal_on_one_gpu(workload, extra):
    """bench.py --rehearse-dist: the N > 1 branch with backend "nccl" (RCCL) and a process group of ONE rank on this box's GPU - init
    with device_id, the barrier, the photon all-gathers on device pointers, the per-frame gather into rank 0's list, the reductions of
    the timing, destroy. Same work as the plain N = 1 line; the line names itself a rehearsal and carries the per-rank figures."""
    common = ["--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu", "--no-counters", "--no-secondary"] + extra
    one = _bench(1, common, env={"MCRT_BENCH_SHARE_GPU": "0"})
    reh = _bench(1, common + ["--rehearse-dist"], env={"MCRT_BENCH_SHARE_GPU": "0"})
    assert "rehearsal" in reh and reh["n_gpus"] == 1
    assert reh["config"]["paths_per_step"] == one["config"]["paths_per_step"]
    assert len(reh["per_rank"]["ms_per_step"]) == 1 and reh["per_rank"]["ms_per_step"][0] > 0 and reh["gather_ms"] > 0
    assert reh["_line"]["per_rank"]["ms_per_step"] and reh["_line"]["gather_ms"] > 0
    if workload == "pm":
        assert reh["photon_allgather_ms"] > 0
        assert reh["config"]["photon_pass"]["global_photons"] == one["config"]["photon_pass"]["global_photons"]
        assert abs(reh["config"]["frame_mean_radiance"] - one["config"]["frame_mean_radiance"]) <= 1e-9 * abs(one["config"]["frame_mean_radiance"])
    else:
        assert reh["config"]["rays_per_step"] == one["config"]["rays_per_step"]
        assert reh["config"]["frame_mean_radiance"] == one["config"]["frame_mean_radiance"]
