"""Checks the product's per-lane device code (monte-carlo-ray-tracer_amd/csrc/*.hpp — the functions the
gfx950 kernels inline) compiled for the host (tests/emu) against the reference's golden dumps and
the oracle. This runs in the GPU-less container; the GPU execution of the same code through the
C ABI is checked in test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import camera_for, check_hits_against_reference, golden_path, load_radiance, rel_error


def _emu_render(emu, pkg, img, cam, seed, integ, rows, stage):
    r0, r1 = rows
    out = np.zeros((r1 - r0, cam.width, 3))
    cnt = (C.c_uint64 * 5)()
    g, c = img.photons(0), img.photons(1)
    rc = emu.emu_render(C.byref(img.scene), C.byref(g) if g is not None else None, C.byref(c) if c is not None else None,
                        img.param("k_nearest_photons") or 50, int(img.param("direct_visualization")), C.byref(cam), seed,
                        integ, r0, r1, stage, out.ctypes.data, cnt)
    assert rc == 0
    return out, dict(rays=cnt[0], node_tests=cnt[1], prim_tests=cnt[2], overflow=cnt[3], paths=cnt[4])


@pytest.mark.parametrize("name,stage", [("hexagon_room_diffuse", 1), ("hexagon_room", 0), ("hexagon_room_ggx", 2),
                                        ("hexagon_room", 2), ("ior_test", 2), ("veach_mis", 2), ("hexagon_room_dof", 2), ("hexagon_room_dof", 0),
                                        ("coffee_maker_qsah", 1), ("coffee_maker_bsah", 0), ("ior_test", 1),
                                        ("veach_mis", 1), ("metals", 0), ("oren_nayar_test", 1), ("ggx_test", 0),
                                        ("quadric", 0), ("quadric", 1)])
def test_path_tracer_device_code_equals_reference(pkg, emu, oracle, manifest, name, stage):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    out, cnt = _emu_render(emu, pkg, img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, r["rows"], stage)
    ref = load_radiance(r)
    # depth-first traversal instead of the reference's best-first heap: same closest hits -> same bits
    assert np.array_equal(out, ref), "max rel err %.3e" % rel_error(out, ref).max()
    assert cnt["overflow"] == 0
    # same rays as the reference-equivalent oracle (one Scene::intersect call per bounce/shadow ray)
    _, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert cnt["rays"] == info["rays"] and cnt["paths"] == info["paths"]


@pytest.mark.parametrize("name,stage_all", [("hexagon_room", 1), ("hexagon_room_ggx", 0), ("coffee_maker_qsah", 0),
                                            ("coffee_maker_bsah", 0), ("veach_mis", 1), ("metals", 0), ("ggx_test", 1),
                                            ("quadric", 0), ("quadric", 1)])
def test_lane_state_machine_equals_reference(pkg, emu, oracle, manifest, name, stage_all):
    """mcrt_lanesm.hpp (the kernel used for every scene whose BVH is walked): split next-event
    estimate, step-wise traversal with packed stack entries — same bits as the reference."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    r0, r1 = r["rows"]
    out = np.zeros((r1 - r0, cam.width, 3))
    cnt = (C.c_uint64 * 5)()
    rc = emu.emu_render_sm(C.byref(img.scene), C.byref(cam), manifest["seed"], r0, r1, stage_all, out.ctypes.data, cnt)
    assert rc == 0 and cnt[3] == 0
    ref = load_radiance(r)
    assert np.array_equal(out, ref), "max rel err %.3e" % rel_error(out, ref).max()
    _, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert cnt[4] == info["paths"]
    # shadow rays whose BSDF term is zero are not traced (the reference traces them and then discards them)
    assert 0 <= info["rays"] - cnt[0] <= 0.03 * info["rays"]


@pytest.mark.parametrize("name,slots", [("hexagon_room", 1000), ("hexagon_room_ggx", 64), ("coffee_maker_qsah", 4096),
                                        ("coffee_maker_bsah", 7), ("veach_mis", 100000), ("metals", 333), ("hexagon_room_dof", 512),
                                        ("quadric", 300)])
def test_wavefront_equals_reference(pkg, emu, oracle, manifest, name, slots):
    """mcrt_wavefront.hpp (path state pooled in HBM, one shade pass + one trace pass per bounce): any number
    of slots — fewer than pixels, more than pixels, not a multiple of anything — gives the reference's bits."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    assert r["rows"] == [0, cam.height]
    out = np.zeros((cam.height, cam.width, 3))
    cnt = (C.c_uint64 * 6)()
    rc = emu.emu_render_wf(C.byref(img.scene), C.byref(cam), manifest["seed"], slots, cam.height, out.ctypes.data, cnt)
    assert rc == 0 and cnt[3] == 0
    ref = load_radiance(r)
    assert np.array_equal(out, ref), "max rel err %.3e" % rel_error(out, ref).max()
    _, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert cnt[4] == info["paths"]
    assert 0 <= info["rays"] - cnt[0] <= 0.03 * info["rays"]


def test_wavefront_sharded_rows(pkg, emu, manifest):
    """Rows dealt to shards: the union of two shards' packed rows is the full frame."""
    case = manifest["cases"]["metals"]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    ref = load_radiance(r)
    for index in (0, 1):
        cam = camera_for(img, r)
        cam.shard_index, cam.shard_count, cam.shard_rows = index, 2, 4
        rows = pkg.shard_rows(cam)
        out = np.zeros((len(rows), cam.width, 3))
        cnt = (C.c_uint64 * 6)()
        assert emu.emu_render_wf(C.byref(img.scene), C.byref(cam), manifest["seed"], 200, len(rows), out.ctypes.data, cnt) == 0
        assert np.array_equal(out, ref[rows])


@pytest.mark.parametrize("slots", [500, 100000])
def test_wavefront_photon_mapper_device_code(pkg, emu, manifest, slots):
    """The photon mapper in wavefront form (estimate requests, a kNN pass, per-lane estimate sums in the next shade pass)."""
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    out = np.zeros((cam.height, cam.width, 3))
    cnt = (C.c_uint64 * 7)()
    g, c = img.photons(0), img.photons(1)
    rc = emu.emu_render_wf_pm(C.byref(img.scene), C.byref(g), C.byref(c), img.param("k_nearest_photons") or 50,
                              int(img.param("direct_visualization")), C.byref(cam), manifest["seed"], slots, cam.height, out.ctypes.data, cnt)
    assert rc == 0 and cnt[3] == 0 and cnt[6] > 0
    # the k photons are summed in heap-array order, which differs from the reference's -> few ulp
    assert rel_error(out, load_radiance(r)).max() < 1e-12


def test_photon_mapper_device_code(pkg, emu, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    out, _ = _emu_render(emu, pkg, img, camera_for(img, r), manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER, r["rows"], 1)
    ref = load_radiance(r)
    # the k photons are summed in heap-array order, which differs from the reference's -> few ulp
    assert rel_error(out, ref).max() < 1e-12


def test_sampler_byte_tables_equal_reference(emu, manifest):
    d = golden_path(manifest["cases"]["hexagon_room"]["kat"])
    inp = np.fromfile(os.path.join(d, "sampler_in.u32"), dtype=np.uint32).reshape(-1, 3)
    ref = np.fromfile(os.path.join(d, "sampler_out.f64")).reshape(-1, 7)
    out = np.empty(7)
    for (pixel, index, shuffles), want in zip(inp, ref):
        emu.emu_sampler(manifest["seed"], int(pixel), int(index), int(shuffles), out.ctypes.data)
        np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize("name", ["hexagon_room", "coffee_maker_qsah", "ior_test", "quadric"])
def test_traversal_device_code_kat(pkg, emu, oracle, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    d = golden_path(case["kat"])
    rays = np.fromfile(os.path.join(d, "isect_rays.f64")).reshape(-1, 6)
    n = rays.shape[0]
    start, direction = rays[:, :3].copy(), rays[:, 3:].copy()
    results = []
    # top-of-tree staged / whole scene staged / flat loop / quantised child blocks of the trace kernel
    stages = (0, 1, 2, 3) if img.scene.num_nodes else (0, 1, 2)
    if img.scene.num_quadrics:
        stages = (0, 1, 3)  # the flat loop knows triangles and spheres only
    for stage in stages:
        t, surf, uv = np.empty(n), np.empty(n, dtype=np.uint32), np.empty((n, 2))
        rc = emu.emu_intersect(C.byref(img.scene), n, start.ctypes.data, direction.ctypes.data, stage, t.ctypes.data,
                               surf.ctypes.data, uv.ctypes.data)
        assert rc == 0
        check_hits_against_reference(oracle, img, d, t, surf, uv)
        results.append((t, surf, uv))
    # The traversal flavours agree bit for bit, ties included — except for rays with an exactly
    # zero direction component, where a slab product can be 0*inf = NaN and BoundingBox::intersect
    # (bounding-box.cpp:9-17) rejects a box that does contain a hit; there the BVH walk follows the
    # reference's BVH result and the flat loop follows the reference's brute-force result.
    generic = np.all(direction != 0.0, axis=1)
    for r in results[1:]:
        for a, b in zip(results[0], r):
            np.testing.assert_array_equal(a[generic], b[generic])


def test_knn_device_code_kat(pkg, emu, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    k = img.param("k_nearest_photons")
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)
        n = pts.shape[0]
        cnt, idx, d2 = np.empty(n, dtype=np.uint32), np.empty((n, k), dtype=np.uint32), np.empty((n, k))
        emu.emu_knn(C.byref(img.photons(which)), n, pts.ctypes.data, k, cnt.ctypes.data, idx.ctypes.data, d2.ctypes.data)
        np.testing.assert_array_equal(cnt, np.fromfile(os.path.join(d, "knn_%s_count.u32" % tag), dtype=np.uint32))
        np.testing.assert_array_equal(d2, np.fromfile(os.path.join(d, "knn_%s_d2.f64" % tag)).reshape(-1, k))
        np.testing.assert_array_equal(idx, np.fromfile(os.path.join(d, "knn_%s_index.u32" % tag), dtype=np.uint32).reshape(-1, k))
