"""The wave-cooperative photon search of the product (csrc/mcrt_waveknn.hpp) on the HOST: the device source unchanged, one wavefront
emulated as 64 fibers that meet at every cross-lane operation (tests/emu/wave_emu.hpp: ballots, readlane, DPP moves with their row
masks and bound_ctrl, shuffles, wave barriers; LDS as ordinary memory). What the GPU tier checks through the C ABI
(test_knn_exact, test_knn_large_k) is checked here without a GPU: the reference's own k-NN vectors, the oracle's search for other k,
both widths of the candidate buffer, and the frontier's spill list in both of its forms (state in registers / in LDS)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import golden_path


def _search(wave_emu, desc, pts, k, rows, spill_mode, upload_k=50):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n = len(pts)
    cnt = np.zeros(n, dtype=np.uint32)
    idx = np.zeros((n, k), dtype=np.uint32)
    d2 = np.zeros((n, k))
    overflow = np.zeros(1, dtype=np.uint32)
    rc = wave_emu.wemu_knn(C.byref(desc), upload_k, n, pts.ctypes.data, k, rows, spill_mode, cnt.ctypes.data, idx.ctypes.data, d2.ctypes.data,
                           overflow.ctypes.data)
    assert rc == 0
    return cnt, idx, d2, int(overflow[0])


def test_wave_search_on_the_host_gives_the_reference_vectors(pkg, wave_emu, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    k = img.param("k_nearest_photons")
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)[:400]
        cnt, idx, d2, overflow = _search(wave_emu, img.photons(which), pts, k, 4, 1, upload_k=k)
        assert overflow == 0
        np.testing.assert_array_equal(cnt, np.fromfile(os.path.join(d, "knn_%s_count.u32" % tag), dtype=np.uint32)[:400])
        np.testing.assert_array_equal(idx, np.fromfile(os.path.join(d, "knn_%s_index.u32" % tag), dtype=np.uint32).reshape(-1, k)[:400])
        np.testing.assert_array_equal(d2, np.fromfile(os.path.join(d, "knn_%s_d2.f64" % tag)).reshape(-1, k)[:400])


@pytest.mark.parametrize("k,rows", [(1, 4), (7, 4), (64, 4), (128, 4), (129, 16), (300, 16), (768, 16), (50, 16)])
def test_wave_search_on_the_host_equals_the_oracle(pkg, wave_emu, oracle, manifest, k, rows):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)[400:440]
        for mode in (1, 2):  # the spill list's state in registers / in LDS (unused here: same answers)
            cnt, idx, d2, overflow = _search(wave_emu, img.photons(which), pts, k, rows, mode)
            ocnt, oidx, od2 = oracle.knn(img.photons(which), pts, k)
            assert overflow == 0
            np.testing.assert_array_equal(cnt, ocnt)
            np.testing.assert_array_equal(d2, od2)
            np.testing.assert_array_equal(idx, oidx)


@pytest.mark.parametrize("count", [1, 49, 50, 51, 300])
def test_wave_search_on_the_host_small_maps(pkg, wave_emu, count):
    rng = np.random.default_rng(count)
    lo, hi = np.array([-3.0, -2.0, -1.0]), np.array([5.0, 2.0, 4.0])
    ph = np.zeros((count, 8), dtype=np.float32)
    ph[:, 3:6] = (lo + rng.random((count, 3)) * (hi - lo)).astype(np.float32)
    m = pkg.PhotonMap(ph, lo.tolist(), hi.tolist(), 200)
    pts = lo + rng.random((40, 3)) * (hi - lo)
    pos = np.ctypeslib.as_array(m.desc.photons, (count, 8))[:, 3:6].astype(np.float64)
    dd = pts[:, None, :] - pos[None, :, :]
    d2_all = (dd[:, :, 0] * dd[:, :, 0] + dd[:, :, 1] * dd[:, :, 1]) + dd[:, :, 2] * dd[:, :, 2]
    for k in (1, 50):
        want = np.sort(d2_all, axis=1)[:, :min(k, count)]
        cnt, idx, d2, overflow = _search(wave_emu, m.desc, pts, k, 4, 1)
        assert overflow == 0 and np.all(cnt == min(k, count))
        np.testing.assert_array_equal(d2[:, :min(k, count)], want)
        assert np.all(np.isinf(d2[:, min(k, count):])) and np.all(idx[:, min(k, count):] == 0xFFFFFFFF)
    m.close()


@pytest.mark.parametrize("leaf,k,rows,upload_k", [(1, 64, 4, 1), (2, 100, 4, 2), (4, 300, 16, 16)])
def test_frontier_spill_list_on_the_host(pkg, wave_emu, leaf, k, rows, upload_k):
    """Leaves far smaller than k, and record lists built for a far smaller k than the one asked for (upload_k: only octants of up to
    that many photons are scanned whole): more octants lie within the bound at once than the 128 frontier entries of a wave's
    registers. Without the list in memory the search reports an overflow; with it - its state kept in registers or in LDS - the
    brute-force answer."""
    rng = np.random.default_rng(leaf * 1000 + k)
    count = 6000
    lo, hi = np.array([-3.0, -2.0, -1.0]), np.array([5.0, 2.0, 4.0])
    ph = np.zeros((count, 8), dtype=np.float32)
    ph[:, 3:6] = (lo + rng.random((count, 3)) * (hi - lo)).astype(np.float32)
    m = pkg.PhotonMap(ph, lo.tolist(), hi.tolist(), leaf)
    pts = lo + rng.random((24, 3)) * (hi - lo)
    pos = np.ctypeslib.as_array(m.desc.photons, (count, 8))[:, 3:6].astype(np.float64)
    dd = pts[:, None, :] - pos[None, :, :]
    d2_all = (dd[:, :, 0] * dd[:, :, 0] + dd[:, :, 1] * dd[:, :, 1]) + dd[:, :, 2] * dd[:, :, 2]
    want = np.sort(d2_all, axis=1)[:, :k]
    _, _, _, overflow = _search(wave_emu, m.desc, pts, k, rows, 0, upload_k=upload_k)
    assert overflow == 1  # the case is a real one: 128 entries do not hold this frontier
    for mode in (1, 2):
        cnt, idx, d2, overflow = _search(wave_emu, m.desc, pts, k, rows, mode, upload_k=upload_k)
        assert overflow == 0 and np.all(cnt == k)
        np.testing.assert_array_equal(d2, want)
        np.testing.assert_array_equal(d2_all[np.arange(len(pts))[:, None], idx], want)
    m.close()


@pytest.mark.parametrize("name", ["coffee_maker_qsah", "coffee_maker_bsah", "hexagon_room", "quadric", "metals"])
def test_shared_leaf_walk_on_the_host_equals_the_oracle(pkg, wave_walk_emu, oracle, manifest, name):
    """traceWalkShared (csrc/mcrt_sharedleaf.hpp: the trace kernel's walk - deferred leaves tested by the whole wave, items numbered by
    a prefix sum and pulled with ds_bpermute, one pop site, the stack's top cached in registers) and traceWalkQ, 64 rays per emulated
    wavefront, full waves and waves with lanes that carry no ray: t, surface and uv of the oracle's Scene::intersect bit for bit,
    on the reference's KAT rays and on random rays (some with zero direction components: the exact-record walk)."""
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    img = pkg.SceneImage(golden_path(case["image"]))
    sc = img.scene
    rng = np.random.default_rng(11)
    lo, hi = np.array(sc.bb_min[:]), np.array(sc.bb_max[:])
    n = 1500
    start = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:40, rng.integers(0, 3)] = 0.0
    d[:40] /= np.linalg.norm(d[:40], axis=1, keepdims=True)
    kat = os.path.join(golden_path(case["kat"]), "isect_rays.f64") if case.get("kat") else ""
    if kat and os.path.exists(kat):
        rays = np.fromfile(kat).reshape(-1, 6)[:1500]
        start, d = np.vstack([start, rays[:, :3]]), np.vstack([d, rays[:, 3:]])
    start, d = np.ascontiguousarray(start), np.ascontiguousarray(d)
    n = start.shape[0]
    t0, s0, uv0, _ = oracle.intersect(img, start, d)
    for which, holes in ((0, 0), (0, 3), (1, 0)):
        t, surf, uv = np.full(n, np.nan), np.zeros(n, dtype=np.uint32), np.zeros((n, 2))
        rc = wave_walk_emu.wemu_intersect(C.byref(sc), n, start.ctypes.data, d.ctypes.data, which, holes, t.ctypes.data, surf.ctypes.data, uv.ctypes.data)
        assert rc == 0
        hit = s0 != 0xFFFFFFFF
        np.testing.assert_array_equal(surf, s0, err_msg="walk %d holes %d" % (which, holes))
        np.testing.assert_array_equal(t[hit], t0[hit])
        np.testing.assert_array_equal(uv[hit], uv0[hit])


@pytest.mark.parametrize("k,rows", [(50, 4), (200, 16)])
def test_wave_radiance_estimate_on_the_host(pkg, wave_walk_emu, manifest, k, rows):
    """renderKernelPM's part 2 on the emulated wavefront - the asking lanes' Interactions staged in memory, one search per request by
    the whole wave, the record read back into 'scalar registers' (readfirstlane), the k photons evaluated by k lanes, the sums by the
    DPP reduction (waveSumD) - against the per-lane estimate functions of the legacy kernel (estimateCausticRadiance /
    estimateGlobalRadiance) at the first hits of a small frame's camera rays. Same photons, same per-photon arithmetic; only the
    order of the FP64 sum differs: 1e-12."""
    from conftest import camera_for
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 32, 18, 1
    n = cam.width * cam.height
    lane, wave = np.zeros((n, 6)), np.zeros((n, 6))
    valid = np.zeros(n, dtype=np.uint8)
    rc = wave_walk_emu.wemu_estimate(C.byref(img.scene), C.byref(img.photons(0)), C.byref(img.photons(1)), k, rows, C.byref(cam),
                                     manifest["seed"], n, lane.ctypes.data, wave.ctypes.data, valid.ctypes.data)
    assert rc == 0
    v = valid != 0
    assert v.sum() > n // 2 and (lane[v] != 0).any()
    rel = np.abs(wave[v] - lane[v]) / np.maximum(np.abs(lane[v]), 1e-3)
    print("estimates at %d hits, k = %d: max rel %.3e between the wave's and the lane's sums" % (v.sum(), k, rel.max()))
    assert rel.max() <= 1e-12
    assert (wave[~v] == 0).all()


def _random_rays(sc, n, seed, kat_dir=None):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(sc.bb_min[:]), np.array(sc.bb_max[:])
    start = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:40, rng.integers(0, 3)] = 0.0   # zero direction components: the exact-record walk
    d[:40] /= np.linalg.norm(d[:40], axis=1, keepdims=True)
    if kat_dir and os.path.exists(os.path.join(kat_dir, "isect_rays.f64")):
        rays = np.fromfile(os.path.join(kat_dir, "isect_rays.f64")).reshape(-1, 6)[:2000]
        start, d = np.vstack([start, rays[:, :3]]), np.vstack([d, rays[:, 3:]])
    return np.ascontiguousarray(start), np.ascontiguousarray(d)


# (form, workgroups, waves per workgroup, stack entries per lane in LDS, refill gate, leaf gate, log2 of the dealing block)
# form 3: round 4's visit (what MCRT_COUNT_TESTS and MCRT_WF_LEAN=0 run);
# forms 11 / 27 (round 5): the shared form with the lean visit (MCRT_WF_LEAN: FP32 ray kept, the pushes as one block of LDS writes) / ... one
# block per visit (trees without a node of more than four children; skipped otherwise). Stacks of 4 LDS entries make lanes leave the fast pushes.
TRACE_LAUNCHES = [(3, 3, 2, 16, 16, 16, 6), (3, 1, 4, 4, 16, 16, 6), (3, 2, 2, 16, 1, 1, 6), (3, 2, 3, 16, 48, 40, 7), (3, 5, 1, 8, 16, 16, 6),
                  (11, 3, 2, 16, 16, 16, 6), (11, 1, 4, 4, 16, 16, 6), (11, 2, 2, 3, 1, 1, 6), (11, 2, 3, 16, 48, 40, 7), (11, 5, 1, 8, 16, 16, 6),
                  (27, 3, 2, 16, 16, 16, 6), (27, 1, 4, 4, 16, 16, 6), (27, 2, 2, 3, 1, 1, 6), (27, 2, 3, 16, 48, 40, 7), (27, 5, 1, 8, 16, 16, 6)]


@pytest.mark.parametrize("name", ["coffee_maker_qsah", "coffee_maker_bsah", "hexagon_room", "quadric", "metals", "veach_mis"])
def test_trace_kernel_on_the_host_equals_the_oracle(pkg, wave_kernel_emu, oracle, manifest, name):
    """wfTraceKernel - the kernel half the bench's GPU time is spent in - as it is, on emulated workgroups: every form (shared leaf
    step = the default, deferred leaves, the first walk, eight-wide nodes), several launch shapes (one to five workgroups of one to
    four waves, 4 to 16 stack entries per lane in LDS with the rest spilled, refill and leaf gates from 'at once' to 'almost never',
    dealing blocks of 64 and 128): t, surface and uv of the oracle's Scene::intersect for every ray, bit for bit, and the kernel's own
    ray count. What this covers beyond the per-wave walk test: the queue dealt in blocks, the LDS cursor, batched refills and hit
    stores, the staging of the tree's top and root, __syncthreads."""
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    img = pkg.SceneImage(golden_path(case["image"]))
    sc = img.scene
    start, d = _random_rays(sc, 3000, 23, golden_path(case["kat"]) if case.get("kat") else None)
    n = len(start)
    t0, s0, uv0, _ = oracle.intersect(img, start, d)
    hit = s0 != 0xFFFFFFFF
    ran = set()
    for form, grid, waves, lds_stack, refill, leaf, deal in TRACE_LAUNCHES:
        t, surf, uv = np.full(n, np.nan), np.full(n, 7, dtype=np.uint32), np.zeros((n, 2))
        stats = np.zeros(64, dtype=np.uint64)
        rc = wave_kernel_emu.wemu_trace_kernel(C.byref(sc), n, start.ctypes.data, d.ctypes.data, form, grid, waves, 0xFFFFFFFF, lds_stack, refill, leaf,
                                               deal, t.ctypes.data, surf.ctypes.data, uv.ctypes.data, stats.ctypes.data)
        if rc == -201 or (rc == -203 and form == 27):
            continue  # (no eight-wide nodes for this tree / a node with more than four children)
        ran.add(form)
        what = "form %d, %d x %d waves, stack %d, gates %d / %d" % (form, grid, waves, lds_stack, refill, leaf)
        assert rc == 0, what
        assert int(stats[1]) == n, what
        np.testing.assert_array_equal(surf, s0, err_msg=what)
        np.testing.assert_array_equal(t[hit], t0[hit], err_msg=what)
        np.testing.assert_array_equal(uv[hit], uv0[hit], err_msg=what)
    assert {3, 11} <= ran and (27 in ran or name != "coffee_maker_qsah")  # (a quaternary tree is single-block by construction)
    print("%s: forms run %s" % (name, sorted(ran)))


KERNEL_NAMES = {1: "renderKernel<path tracer, flat>", 2: "renderKernel (wave-synchronous)", 3: "renderKernelSM", 5: "renderKernelPM",
                16: "renderKernelFlatK (flat, cull records as a kernel argument)"}


def _emulated_frame(pkg, wave_kernel_emu, img, cam, seed, integrator, force=0, grid=1):
    out = np.zeros((cam.height, cam.width, 3))
    stats = np.zeros(64, dtype=np.uint64)
    kid = C.c_int(0)
    g, c = img.photons(0), img.photons(1)
    rc = wave_kernel_emu.wemu_render(C.byref(img.scene), C.byref(g) if g is not None else None, C.byref(c) if c is not None else None,
                                     img.param("k_nearest_photons") or 50, int(img.param("direct_visualization") or 0), C.byref(cam), seed, integrator,
                                     force, grid, out.ctypes.data, stats.ctypes.data, C.byref(kid))
    return rc, out, stats, kid.value


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_diffuse", "hexagon_room_ggx", "hexagon_room_dof", "coffee_maker_qsah", "coffee_maker_bsah",
                                  "veach_mis", "metals", "ggx_test", "oren_nayar_test", "ior_test", "quadric", "dragon_room", "shell_room"])
def test_frame_kernels_on_the_host_give_the_oracle_frame(pkg, wave_kernel_emu, oracle, manifest, name):
    """The FRAME KERNELS of the product - the flat megakernel of the headline config, the lane state machine, the wave-synchronous
    kernel - as they are, on emulated workgroups of 8 wavefronts (work units popped with wave-aggregated atomics, path regeneration,
    LDS staging of scene / tables / stacks, the per-sample store) followed by sampleResolveKernel: a small frame of every golden scene,
    by the kernel launchRender picks and by the wave-synchronous one, against the oracle's frame (which is the reference's, bit for
    bit: test_oracle_vs_reference.py) - bit for bit, with the oracle's ray count. No GPU."""
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    from conftest import camera_for
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 28, 16, 2
    want, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    seen = set()
    for force, grid in ((0, 1), (2, 2), (6, 2)):  # (6: the flat megakernel with its cull records as a kernel argument, renderKernelFlatK)
        rc, out, stats, kid = _emulated_frame(pkg, wave_kernel_emu, img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, force, grid)
        if force == 6 and (rc == -206 or kid != 16):
            continue  # (not a flat scene, or more records than the argument block holds)
        assert rc == 0, "%s: rc %d" % (KERNEL_NAMES.get(kid), rc)
        seen.add(kid)
        assert int(stats[0]) == cam.width * cam.height * 4 and int(stats[1]) == info["rays"], KERNEL_NAMES.get(kid)
        np.testing.assert_array_equal(out, want, err_msg="%s: %s is not the oracle's frame" % (name, KERNEL_NAMES.get(kid)))
    print("%s: %s - the oracle's bits, %d rays" % (name, " and ".join(KERNEL_NAMES[k] for k in sorted(seen)), info["rays"]))


def test_photon_mapping_kernel_on_the_host(pkg, wave_kernel_emu, oracle, manifest):
    """renderKernelPM (1024 lanes = 16 emulated wavefronts: per-lane bounce code, estimates served by whole waves) on hexagon_room_pm's
    maps against the oracle's photon-mapped frame: the order of an estimate's FP64 sum is the only difference (1e-12)."""
    from conftest import camera_for
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 24, 12, 2
    want, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    rc, out, stats, kid = _emulated_frame(pkg, wave_kernel_emu, img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert rc == 0 and kid == 5
    assert int(stats[1]) == info["rays"] and int(stats[4]) > 0
    rel = np.abs(out - want) / np.maximum(np.abs(want), 1e-3)
    print("hexagon_room_pm through renderKernelPM on the host: max rel %.3e, %d searches" % (rel.max(), int(stats[4])))
    assert rel.max() <= 1e-12


def _emulated_pipeline_frame(wave_kernel_emu, img, cam, seed, integrator, slots, grid, waves, form):
    out = np.zeros((cam.height, cam.width, 3))
    stats = np.zeros(64, dtype=np.uint64)
    launches = C.c_uint32(0)
    g, c = img.photons(0), img.photons(1)
    rc = wave_kernel_emu.wemu_render_pipeline(C.byref(img.scene), C.byref(g) if g is not None else None, C.byref(c) if c is not None else None,
                                              img.param("k_nearest_photons") or 50, int(img.param("direct_visualization") or 0), C.byref(cam), seed,
                                              integrator, slots, grid, waves, form, out.ctypes.data, stats.ctypes.data, C.byref(launches))
    return rc, out, stats, launches.value


# (pool slots, trace workgroups, waves per trace workgroup, trace kernel form)
PIPELINE_LAUNCHES = [(512, 2, 2, 3), (256, 1, 4, 3), (1024, 3, 1, 3), (768, 2, 2, 11), (512, 2, 2, 11), (256, 1, 4, 27), (1024, 3, 1, 27)]


@pytest.mark.parametrize("name", ["coffee_maker_qsah", "coffee_maker_bsah", "hexagon_room", "hexagon_room_dof", "quadric", "metals", "veach_mis", "ggx_test",
                                  "dragon_room", "shell_room"])
def test_wavefront_pipeline_on_the_host_gives_the_oracle_frame(pkg, wave_kernel_emu, oracle, manifest, name):
    """The wavefront pipeline as launchWavefront runs it - wfShadeKernel (one lane per pool slot: NEE finish, shading, regeneration, rays
    appended to the queue with wave-aggregated atomics under divergent control flow), wfTraceKernel<PoolRays> in every form, launch after
    launch until a shade launch queues nothing, then sampleResolveKernel - on emulated workgroups, the slot pool starting as garbage
    like device memory: the oracle's frame, bit for bit, for several pool sizes and launch shapes. (Its ray count may fall short of the
    oracle's by the shadow rays whose BSDF term is zero: the pipeline does not trace them.)"""
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    from conftest import camera_for
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 24, 14, 2
    want, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    for slots, grid, waves, form in PIPELINE_LAUNCHES:
        rc, out, stats, launches = _emulated_pipeline_frame(wave_kernel_emu, img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, slots, grid, waves, form)
        if rc == -201 or (rc == -204 and form == 27):
            continue  # (no eight-wide nodes for this tree / a node with more than four children)
        what = "%s: %d slots, trace %d x %d waves, form %d" % (name, slots, grid, waves, form)
        assert rc == 0, what
        assert int(stats[0]) == cam.width * cam.height * 4 and 0 <= info["rays"] - int(stats[1]) <= 0.03 * info["rays"], what
        assert launches > 6
        np.testing.assert_array_equal(out, want, err_msg=what)


def test_photon_mapped_pipeline_on_the_host(pkg, wave_kernel_emu, oracle, manifest):
    """... and the photon-mapped pipeline: estimate requests staged by the shade launch, served by wfKnnKernel<eval> (one request per
    wave), read back by the next shade launch."""
    from conftest import camera_for
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 20, 12, 2
    want, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    rc, out, stats, launches = _emulated_pipeline_frame(wave_kernel_emu, img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER, 512, 2, 2, 3)
    assert rc == 0 and int(stats[4]) > 0
    rel = np.abs(out - want) / np.maximum(np.abs(want), 1e-3)
    print("hexagon_room_pm through the pipeline on the host: max rel %.3e, %d searches, %d launches" % (rel.max(), int(stats[4]), launches))
    assert rel.max() <= 1e-12


@pytest.mark.parametrize("name", ["hexagon_room_pm", "coffee_maker_qsah"])
def test_emission_kernel_on_the_host_gives_the_oracle_photons(pkg, wave_kernel_emu, oracle, manifest, name):
    """emitKernel on emulated workgroups (photon paths regenerated from a counter, photons appended with wave-aggregated atomics): the
    oracle's photon lists, record for record by (light, emission, bounce) key - and the sizing pilot (every 64th path, capacity 0)
    counts what it should: within a few per cent of a 64th of the lists."""
    from conftest import sort_by_key
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    emissions, factor = 300, 10.0
    want = oracle.emit_photons(img, emissions, factor, manifest["seed"])
    cap = 1 << 16
    g, gk = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
    c, ck = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
    counts = np.zeros(4, dtype=np.uint64)
    rc = wave_kernel_emu.wemu_emit(C.byref(img.scene), float(emissions), factor, manifest["seed"], 1, 2, cap, g.ctypes.data, gk.ctypes.data, c.ctypes.data,
                                   ck.ctypes.data, counts.ctypes.data)
    assert rc == 0
    assert int(counts[2]) == want["paths"] and int(counts[3]) == want["rays"]
    for (ph, keys, n), (wph, wkeys) in (((g, gk, int(counts[0])), want["global_"]), ((c, ck, int(counts[1])), want["caustic"])):
        assert n == len(wkeys)
        a, ak = sort_by_key(ph[:n], keys[:n])
        np.testing.assert_array_equal(ak, wkeys)
        np.testing.assert_array_equal(a.view(np.uint32), wph.view(np.uint32))
    pilot = np.zeros(4, dtype=np.uint64)
    rc = wave_kernel_emu.wemu_emit(C.byref(img.scene), float(emissions), factor, manifest["seed"], 8, 1, 0, None, None, None, None, pilot.ctypes.data)
    assert rc == 0 and int(pilot[2]) == -(-want["paths"] // 8)
    total, sample = int(counts[0]) + int(counts[1]), (int(pilot[0]) + int(pilot[1])) * 8
    assert abs(sample - total) <= 0.25 * total


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_frames_do_not_depend_on_the_order_the_waves_run_in(pkg, wave_kernel_emu, oracle, manifest, seed):
    """The waves of an emulated workgroup normally take turns; here they are visited in a random order that changes from pass to pass,
    a wave sitting a pass out now and then - arbitration as arbitrary as the hardware's. Work units, queue entries and photon slots
    are then handed out in other orders; frames (flat megakernel, lane state machine, pipeline, photon mapper) and photon sets are not
    allowed to notice."""
    from conftest import camera_for, sort_by_key
    try:
        wave_kernel_emu.wemu_set_shuffle(seed)
        for name, integ in (("hexagon_room", pkg.INTEGRATOR_PATH_TRACER), ("coffee_maker_qsah", pkg.INTEGRATOR_PATH_TRACER),
                            ("hexagon_room_pm", pkg.INTEGRATOR_PHOTON_MAPPER)):
            case = manifest["cases"][name]
            img = pkg.SceneImage(golden_path(case["image"]))
            cam = camera_for(img, case["renders"][0])
            cam.width, cam.height, cam.sqrtspp = 20, 12, 2
            want, info = oracle.render(img, cam, manifest["seed"], integ)
            rc, out, stats, kid = _emulated_frame(pkg, wave_kernel_emu, img, cam, manifest["seed"], integ, 0, 2)
            assert rc == 0
            rc2, pipe, stats2, _ = _emulated_pipeline_frame(wave_kernel_emu, img, cam, manifest["seed"], integ, 512, 2, 3, 3)
            assert rc2 == 0
            if integ == pkg.INTEGRATOR_PATH_TRACER:
                np.testing.assert_array_equal(out, want, err_msg=name)
                np.testing.assert_array_equal(pipe, want, err_msg=name + " (pipeline)")
            else:
                assert (np.abs(out - want) / np.maximum(np.abs(want), 1e-3)).max() <= 1e-12
                assert (np.abs(pipe - want) / np.maximum(np.abs(want), 1e-3)).max() <= 1e-12
    finally:
        wave_kernel_emu.wemu_set_shuffle(0)


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_large_scene_kernels_on_the_host(pkg, wave_kernel_emu, oracle, name):
    """The kernels of the scaled BASELINE configurations on their own trees, without a GPU: C3 (metal_bunnies stand-in, 491 592 triangles:
    lane state machine and the wavefront pipeline) and C5 (water_caustics stand-in, 6 898 815 triangles, photon-mapped: renderKernelPM's
    instance for trees in memory - per-lane stacks partly in LDS, shared-leaf walks, the spill list's state in LDS - and the photon-
    mapped pipeline) render a small frame of the config's camera on emulated workgroups: the oracle's frame (bit for bit path-traced;
    1e-12 photon-mapped: sum order). The images are made by integration/large_scenes/make_large.py (skipped where they are not)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.ensure_image(name)
    if p is None:
        pytest.skip("oracle/_ref (reference binary + scene copies) not on this machine")
    img = pkg.SceneImage(p)
    cam = img.camera
    cam.width, cam.height, cam.sqrtspp = 20, 12, 1
    photon = bool(make_large.CONFIGS[name]["photon"])
    integ = pkg.INTEGRATOR_PHOTON_MAPPER if photon else pkg.INTEGRATOR_PATH_TRACER
    want, info = oracle.render(img, cam, 0x12345678, integ)
    rc, out, stats, kid = _emulated_frame(pkg, wave_kernel_emu, img, cam, 0x12345678, integ, 0, 1)
    assert rc == 0 and kid == (5 if photon else 3) and int(stats[1]) == info["rays"]
    # (the pipeline with the trace form the device runs on THIS tree: C3 is quaternary -> the lean visit, one block per node; C5's
    # octree hierarchy has nodes of up to eight children -> the lean visit with the block loop)
    rc2, pipe, stats2, launches = _emulated_pipeline_frame(wave_kernel_emu, img, cam, 0x12345678, integ, 256, 2, 2, 27 if name == "c3" else 11)
    assert rc2 == 0 and 0 <= info["rays"] - int(stats2[1]) <= 0.03 * info["rays"] + 1
    if photon:
        for got in (out, pipe):
            assert (np.abs(got - want) / np.maximum(np.abs(want), 1e-3)).max() <= 1e-12
    else:
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(pipe, want)


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_dof", "ior_test", "coffee_maker_qsah", "coffee_maker_bsah"])
def test_lean_kernel_instances_on_the_host_give_the_oracle_frame(pkg, wave_kernel_emu_lean, oracle, manifest, name):
    """csrc/mcrt_hip_lean.hip compiles the default path's kernels WITHOUT the rough-diffuse, rough-specular and conductor branches
    (MCRT_MAT_FEATURES_OFF, csrc/mcrt_shade.hpp), for scenes none of whose materials carries one of those flags. The same kernels built
    that way for the host: the megakernel launchRender picks, the flat megakernel with its records as a kernel argument, and the wavefront
    pipeline render such scenes to the oracle's frame, bit for bit. (On the GPU: tests/test_gpu_lean_kernels.py.)"""
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    from conftest import camera_for
    img = pkg.SceneImage(golden_path(case["image"]))
    s = img.scene
    assert not any(s.materials[i].flags & 67 for i in range(s.num_materials)), "the lean instances are not for this scene"
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 24, 14, 2
    want, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    for force, grid in ((0, 1), (6, 2)):
        rc, out, stats, kid = _emulated_frame(pkg, wave_kernel_emu_lean, img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, force, grid)
        if force == 6 and (rc == -206 or kid != 16):
            continue
        assert rc == 0 and int(stats[1]) == info["rays"]
        np.testing.assert_array_equal(out, want, err_msg="%s: lean %s is not the oracle's frame" % (name, KERNEL_NAMES.get(kid)))
    rc, out, stats, launches = _emulated_pipeline_frame(wave_kernel_emu_lean, img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, 512, 2, 2, 11)
    assert rc == 0
    np.testing.assert_array_equal(out, want, err_msg="%s: lean pipeline" % name)


def test_lean_photon_mapping_kernel_on_the_host(pkg, wave_kernel_emu, wave_kernel_emu_lean, manifest):
    """... and renderKernelPM: the lean instance's hexagon_room_pm frame is the full instance's, bit for bit."""
    from conftest import camera_for
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 20, 10, 2
    rc0, full, st0, kid0 = _emulated_frame(pkg, wave_kernel_emu, img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    rc1, lean, st1, kid1 = _emulated_frame(pkg, wave_kernel_emu_lean, img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert rc0 == 0 and rc1 == 0 and kid0 == kid1 == 5 and int(st0[4]) == int(st1[4]) > 0
    np.testing.assert_array_equal(lean, full)
