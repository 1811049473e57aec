"""Parity tests proper: the HIP path on a real MI355X, called through the C ABI (libmcrt_hip.so),
against the reference's golden dumps (tests/golden/, made by the reference itself) and the oracle.

Tolerance (BASELINE.json north_star): per-pixel radiance within 1e-4 relative of the CPU reference at
fixed seed, measured as |gpu-ref| / max(|ref|, 1e-3) per channel. Integer/index work (sampler bits,
hit surface indices, kNN photon indices) must be exact.

Round 3: the device computes the reference's sin / cos pairs with glibc's own sincos algorithm (csrc/mcrt_libm.hpp); round 4: and
Scene::skyColor's asin with glibc's own asin (refAsin) - the last libm call of the path-traced path. EVERY path-traced golden frame
is now required to be the reference's bits (EXACT below = all of them), and the outlier allowance is gone. Photon-mapped
frames keep their own bar AGAINST THE REFERENCE (the k photons of an estimate are summed by a wave reduction, not in heap order:
1e-12); among themselves — passes, chunking, shards, contexts, megakernel against pipeline — they are bit-equal. Round 5: through the
per-lane kernel, which keeps the reference's heap discipline, the photon-mapped frame IS the reference's bits."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import camera_for, check_hits_against_reference, golden_path, load_radiance, rel_error

pytestmark = pytest.mark.gpu
TOL = 1e-4
SMOOTH_TOL = 1e-12  # (photon-mapped frames and film-filter frames: sums in another order)
# every operation of a path is reproduced bit for bit - with or without a sky (refAsin, round 4)
EXACT = {"hexagon_room_diffuse", "hexagon_room", "hexagon_room_ggx", "hexagon_room_dof", "coffee_maker_qsah", "coffee_maker_bsah", "shell_room",
         "dragon_room", "ior_test", "veach_mis", "metals", "oren_nayar_test", "ggx_test", "quadric"}


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def expected_kernel(pkg, img, integrator, forced=None):
    """Which kernel form launchRender must pick (include/mcrt.h MCRT_KERNEL_*): every parity test pins it, so a
    frame that passes on another form than the one the test is about (a silent fallback) fails here."""
    s = img.scene
    photon = integrator == pkg.INTEGRATOR_PHOTON_MAPPER
    kinds = np.ctypeslib.as_array(s.surf_kind, shape=(s.num_surfaces,))
    quadrics = bool((kinds == 2).any())
    flat = s.num_surfaces <= 64 and not quadrics
    if forced == "wf":
        return pkg.KERNEL_WAVEFRONT_PM if photon else pkg.KERNEL_WAVEFRONT
    if photon:
        return pkg.KERNEL_PM_LANE if forced == "legacy" else pkg.KERNEL_PM_WAVE
    if flat and forced in (None, "sm", "legacy"):
        return pkg.KERNEL_FLAT
    if forced == "legacy":
        return pkg.KERNEL_WAVESYNC
    if s.num_nodes == 0:
        return pkg.KERNEL_WAVESYNC  # brute-force Scene::intersect (no BVH) of a scene too large for the flat loop
    if forced is None and s.num_nodes >= 65536:
        return pkg.KERNEL_WAVEFRONT
    return pkg.KERNEL_LANE_SM


def _check(out, ref, what, exact=False, tol=SMOOTH_TOL):
    rel = rel_error(out, ref).max(axis=2)
    bad = int((rel > tol).sum())
    print("%s: max rel %.3e, 99.9th pct %.3e, pixels beyond %g: %d / %d" % (what, rel.max(), np.quantile(rel, 0.999), tol, bad, rel.size))
    assert np.isfinite(out).all()
    if exact:
        np.testing.assert_array_equal(out, ref, err_msg="%s: not the reference's bits" % what)
    assert bad == 0, "%s: %d pixels differ by more than %g" % (what, bad, tol)
    return rel


@pytest.mark.parametrize("name", ["hexagon_room_diffuse", "hexagon_room", "hexagon_room_ggx", "hexagon_room_dof", "coffee_maker_qsah",
                                  "coffee_maker_bsah", "ior_test", "veach_mis", "metals", "oren_nayar_test", "ggx_test", "quadric", "shell_room", "dragon_room"])
def test_path_tracer_matches_reference(pkg, ctx, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    for r in case["renders"]:
        cam = camera_for(img, r)
        r0, r1 = r["rows"]
        if (r0, r1) != (0, r["height"]):
            continue  # crops of the full-size frame: test_c2_full_size_frame
        out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        _check(out, load_radiance(r), "%s %s" % (name, r["file"]), exact=name in EXACT)
        assert st["paths"] == r["width"] * r["height"] * r["sqrtspp"] ** 2
        assert st["rays"] >= st["paths"] and st["kernel_launches"] >= 1
        assert st["kernel_id"] == expected_kernel(pkg, img, pkg.INTEGRATOR_PATH_TRACER), pkg.KERNEL_NAMES.get(st["kernel_id"])


@pytest.fixture
def kernel_env():
    """MCRT_KERNEL is read at every launch: wf = wavefront pipeline, sm = lane-state-machine megakernel."""
    old = os.environ.get("MCRT_KERNEL")
    yield lambda v: os.environ.__setitem__("MCRT_KERNEL", v)
    if old is None:
        os.environ.pop("MCRT_KERNEL", None)
    else:
        os.environ["MCRT_KERNEL"] = old


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_ggx", "hexagon_room_dof", "coffee_maker_qsah", "coffee_maker_bsah",
                                  "veach_mis", "metals", "ggx_test", "quadric"])
def test_wavefront_pipeline_matches_reference(pkg, ctx, manifest, kernel_env, name):
    """The wavefront pipeline (mcrt_wavefront.hpp; default for scenes whose BVH stays in HBM) forced onto the
    golden scenes: the reference's radiance, and the megakernel's bits (same per-path arithmetic, same
    per-pixel summation order, whatever the slot count)."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    r = case["renders"][0]
    cam = camera_for(img, r)
    base, st0 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    kernel_env("wf")
    for slots in ("1048576", "4096"):
        os.environ["MCRT_WF_SLOTS"] = slots
        try:
            out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        finally:
            os.environ.pop("MCRT_WF_SLOTS", None)
        _check(out, load_radiance(r), "%s wavefront (%s slots)" % (name, slots), exact=name in EXACT)
        assert st["paths"] == st0["paths"] and st["kernel_launches"] > 2
        assert st["kernel_id"] == pkg.KERNEL_WAVEFRONT and st0["kernel_id"] == expected_kernel(pkg, img, pkg.INTEGRATOR_PATH_TRACER)
        np.testing.assert_array_equal(out, base)


@pytest.mark.parametrize("name", ["coffee_maker_qsah", "coffee_maker_bsah", "quadric"])
def test_optional_trace_kernels_same_frame(pkg, ctx, manifest, kernel_env, name, monkeypatch):
    """The trace kernel's remaining switches - round 4's visit instead of the lean one (MCRT_WF_LEAN=0), the lean visit with the block
    loop kept (MCRT_WF_LEAN=2), and the leaf gate at both extremes (a shared leaf step for every single pending lane: four item lanes
    per leaf; only when 40 lanes wait: one item lane per leaf) - give the default kernel's frame, bit for bit. (The optional kernels of
    rounds 2-4 - slot-scheduled, eight-wide nodes, lanes waiting at their leaves, a pending leaf tested by its own lane, two half
    pools - lost every A/B and were removed in round 6.)"""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    kernel_env("wf")
    ctx.upload_image(img)
    base, st0 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert st0["kernel_id"] == pkg.KERNEL_WAVEFRONT
    for key, value in (("MCRT_WF_LEAN", "0"), ("MCRT_WF_LEAN", "2"), ("MCRT_WF_LEAF", "1"), ("MCRT_WF_LEAF", "40")):
        monkeypatch.setenv(key, value)
        out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        monkeypatch.delenv(key)
        assert st["rays"] == st0["rays"] and st["kernel_id"] == pkg.KERNEL_WAVEFRONT
        np.testing.assert_array_equal(out, base, err_msg=key)


@pytest.mark.parametrize("name,kernel,integrator", [("hexagon_room", None, "pt"), ("coffee_maker_qsah", None, "pt"), ("metals", "wf", "pt"),
                                                    ("hexagon_room_pm", None, "pm"), ("veach_mis", "legacy", "pt")])
def test_passes_and_chunks_do_not_change_the_frame(pkg, ctx, manifest, kernel_env, name, kernel, integrator):
    """A per-sample store of 100 KB forces the frame through in passes of 8 rows (row offsets of the store, the work
    counter and the resolve), and MCRT_CHUNKS 1 / 4 changes the work units from whole pixels to quarter pixels: the bits of
    the frame must not depend on either (flat instance, state machine, wavefront pipeline, photon kernel, legacy kernel)."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    if integrator == "pm":
        ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons") or 50, bool(img.param("direct_visualization")))
    r = case["renders"][0]
    cam = camera_for(img, r)
    mode = pkg.INTEGRATOR_PHOTON_MAPPER if integrator == "pm" else pkg.INTEGRATOR_PATH_TRACER
    if kernel:
        kernel_env(kernel)
    base, st0 = ctx.sample_image(cam, manifest["seed"], mode)
    _check(base, load_radiance(r), name)
    assert st0["kernel_id"] == expected_kernel(pkg, img, mode, kernel), pkg.KERNEL_NAMES.get(st0["kernel_id"])
    for store, chunks in (("0.0001", None), (None, "1"), ("0.0001", "4")):
        try:
            if store:
                os.environ["MCRT_SAMPLE_STORE_GB"] = store
            if chunks:
                os.environ["MCRT_CHUNKS"] = chunks
            out, st = ctx.sample_image(cam, manifest["seed"], mode)
        finally:
            os.environ.pop("MCRT_SAMPLE_STORE_GB", None)
            os.environ.pop("MCRT_CHUNKS", None)
        assert st["paths"] == st0["paths"] and st["kernel_id"] == st0["kernel_id"]
        if store:
            assert st["kernel_launches"] > st0["kernel_launches"]   # several passes
        # (photon-mapped frames too: an estimate's photons are found and summed in an order that depends on the query alone)
        np.testing.assert_array_equal(out, base)


def test_photon_mapper_matches_reference(pkg, ctx, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    r = case["renders"][0]
    out, st = ctx.sample_image(camera_for(img, r), manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    _check(out, load_radiance(r), "hexagon_room_pm")
    assert st["knn_searches"] > 0 and st["kernel_id"] == pkg.KERNEL_PM_WAVE


def test_photon_mapper_per_lane_kernel_is_the_references_bits(pkg, ctx, manifest, kernel_env):
    """The per-lane photon-mapper kernel (MCRT_KERNEL=legacy: renderKernel<photon_mapper>, every lane its own search) keeps the
    reference's heap discipline - push_unordered up to k - 1 results, make_heap at the k-th, pop_push after (linear-octree.cpp:58-79,
    priority-queue.hpp) - and the restated sincosf for Photon::dir, so the k photons of an estimate are summed in the reference's
    order: the photon-mapped frame is the reference's BITS. What separates the wave-cooperative kernels' frames from the
    reference's (1e-12 above) is therefore the order of those sums and nothing else."""
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons") or 50, bool(img.param("direct_visualization")))
    r = case["renders"][0]
    kernel_env("legacy")
    out, st = ctx.sample_image(camera_for(img, r), manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st["kernel_id"] == pkg.KERNEL_PM_LANE and st["knn_searches"] > 0
    _check(out, load_radiance(r), "hexagon_room_pm per-lane kernel", exact=True)


def test_photon_mapper_wavefront_pipeline(pkg, ctx, manifest, kernel_env):
    """The photon-mapped frame through the wavefront pipeline (trace / kNN / shade launches; the default when the BVH
    has 65 536 nodes or more): the reference's radiance, and the megakernel's up to the order of the estimate sums."""
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons") or 50, bool(img.param("direct_visualization")))
    r = case["renders"][0]
    cam = camera_for(img, r)
    base, st0 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    kernel_env("wf")
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    _check(out, load_radiance(r), "hexagon_room_pm wavefront")
    assert st["paths"] == st0["paths"] and st["knn_searches"] == st0["knn_searches"] and st["kernel_launches"] > 3
    assert st0["kernel_id"] == pkg.KERNEL_PM_WAVE and st["kernel_id"] == pkg.KERNEL_WAVEFRONT_PM
    np.testing.assert_array_equal(out, base)  # the pipeline's kNN launch evaluates an estimate exactly as renderKernelPM does


def test_rays_equal_oracle_count(pkg, ctx, oracle, manifest):
    case = manifest["cases"]["hexagon_room_diffuse"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height = 64, 64
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    ref, info = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    _check(out, ref, "64x64 vs oracle")
    # one flipped branch changes a path's ray count; allow a handful
    assert abs(st["rays"] - info["rays"]) <= 50 and st["paths"] == info["paths"]


def test_sampler_bits_exact(pkg, ctx, manifest):
    d = golden_path(manifest["cases"]["hexagon_room"]["kat"])
    inp = np.fromfile(os.path.join(d, "sampler_in.u32"), dtype=np.uint32).reshape(-1, 3)
    ref = np.fromfile(os.path.join(d, "sampler_out.f64")).reshape(-1, 7)
    for shuffles in range(5):
        sel = inp[:, 2] == shuffles
        out = ctx.sampler(inp[sel, 0].copy(), inp[sel, 1].copy(), shuffles, manifest["seed"])
        np.testing.assert_array_equal(out, ref[sel])


def test_bsdf_kat(pkg, ctx, manifest):
    """mcrt_bsdf — Fresnel::dielectric / conductor, GGX::reflection / transmission / visibleMicrofacet / D / Lambda and the
    Oren-Nayar Material::diffuseReflection, the lobes Interaction::BSDF mixes (ray/interaction.cpp:84-153) — on the reference's
    own vectors (kat_hexagon_room/bsdf_*.f64, written by calling the reference functions). +,-,*,/ and sqrt are correctly rounded
    and contraction is off, and the one libm call (visibleMicrofacet's sin / cos pair) is glibc's own algorithm on the device: every
    column must have the reference's bits."""
    d = golden_path(manifest["cases"]["hexagon_room"]["kat"])
    inp = np.fromfile(os.path.join(d, "bsdf_in.f64")).reshape(-1, 11)
    ref = np.fromfile(os.path.join(d, "bsdf_out.f64")).reshape(-1, 18)
    consts = np.fromfile(os.path.join(d, "bsdf_consts.f64"))
    out = ctx.bsdf(inp, consts)
    # every column: +,-,*,/ and sqrt are correctly rounded, contraction is off, and visibleMicrofacet's sin / cos pair (columns 8-10, which
    # D(m), column 11, inherits) is glibc's sincos restated (csrc/mcrt_libm.hpp) since round 3
    np.testing.assert_array_equal(out, ref)
    print("bsdf KAT: %d vectors x 18 columns bit-equal" % len(inp))


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_diffuse", "coffee_maker_qsah", "ior_test", "quadric"])
def test_intersect_exact(pkg, ctx, oracle, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    d = golden_path(case["kat"])
    rays = np.fromfile(os.path.join(d, "isect_rays.f64")).reshape(-1, 6)
    t, surf, uv = ctx.intersect(rays[:, :3].copy(), rays[:, 3:].copy())
    # FP64 +,-,*,/ and sqrt are correctly rounded on gfx950 and contraction is off: identical bits
    ties = check_hits_against_reference(oracle, img, d, t, surf, uv)
    print("%s: %d exact-t ties resolved to the lowest index" % (name, ties))


def test_flat_and_bvh_modes_agree(pkg, oracle, manifest):
    """Tiny scenes use the wave-uniform flat loop (MCRT_FLAT_MAX, default 64 primitives); the BVH walk of
    the same scene must give the same bits (hits, ties included, and whole frames)."""
    case = manifest["cases"]["hexagon_room"]
    img = pkg.SceneImage(golden_path(case["image"]))
    d = golden_path(case["kat"])
    rays = np.fromfile(os.path.join(d, "isect_rays.f64")).reshape(-1, 6)
    cam = camera_for(img, case["renders"][0])
    results = []
    for flat_max in ("64", "0"):
        os.environ["MCRT_FLAT_MAX"] = flat_max
        c = pkg.Context(0)
        c.upload_image(img)
        hit = c.intersect(rays[:, :3].copy(), rays[:, 3:].copy())
        frame, st = c.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        results.append((hit, frame, st["rays"]))
        assert st["kernel_id"] == (pkg.KERNEL_FLAT if flat_max == "64" else pkg.KERNEL_LANE_SM)
        c.close()
    del os.environ["MCRT_FLAT_MAX"]
    (h0, f0, r0), (h1, f1, r1) = results
    generic = np.all(rays[:, 3:] != 0.0, axis=1)  # a zero direction component makes NaN slabs (see emulation test)
    for a, b in zip(h0, h1):
        np.testing.assert_array_equal(a[generic], b[generic])
    assert np.array_equal(f0, f1)
    assert 0 <= r0 - r1 <= 0.03 * r0  # the BVH kernel does not trace shadow rays whose BSDF term is zero
    _check(f0, load_radiance(case["renders"][0]), "hexagon_room flat mode")


def test_knn_exact(pkg, ctx, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    k = img.param("k_nearest_photons")
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)
        cnt, idx, d2 = ctx.knn(which, pts, k)
        np.testing.assert_array_equal(cnt, np.fromfile(os.path.join(d, "knn_%s_count.u32" % tag), dtype=np.uint32))
        np.testing.assert_array_equal(idx, np.fromfile(os.path.join(d, "knn_%s_index.u32" % tag), dtype=np.uint32).reshape(-1, k))
        np.testing.assert_array_equal(d2, np.fromfile(os.path.join(d, "knn_%s_d2.f64" % tag)).reshape(-1, k))


def test_knn_exact_four_queries_per_wave(pkg, ctx, manifest, monkeypatch):
    """mcrt_groupknn.hpp (MCRT_KNN_GROUPS=1: one query per row of 16 lanes, four per wave) against the reference's vectors,
    for the reference's k and for k that leave the buffer nearly empty / nearly full."""
    monkeypatch.setenv("MCRT_KNN_GROUPS", "1")
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    k = img.param("k_nearest_photons")
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)
        cnt, idx, d2 = ctx.knn(which, pts, k)
        np.testing.assert_array_equal(cnt, np.fromfile(os.path.join(d, "knn_%s_count.u32" % tag), dtype=np.uint32))
        np.testing.assert_array_equal(idx, np.fromfile(os.path.join(d, "knn_%s_index.u32" % tag), dtype=np.uint32).reshape(-1, k))
        np.testing.assert_array_equal(d2, np.fromfile(os.path.join(d, "knn_%s_d2.f64" % tag)).reshape(-1, k))
        for kk in (1, 7, 64):  # ... and the same answers as one query per wave
            got = ctx.knn(which, pts[:1000], kk)
            monkeypatch.setenv("MCRT_KNN_GROUPS", "0")
            want = ctx.knn(which, pts[:1000], kk)
            monkeypatch.setenv("MCRT_KNN_GROUPS", "1")
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("count", [1, 7, 49, 50, 51, 300, 5000])
def test_knn_small_and_odd_maps(pkg, ctx, manifest, monkeypatch, count):
    """k-NN on maps with fewer photons than k, exactly k, just more, and a few leaves: one query per wave and four per wave
    against a brute-force selection with the reference's distance expression (glm::distance2 on the float positions)."""
    img = pkg.SceneImage(golden_path(manifest["cases"]["hexagon_room_pm"]["image"]))
    ctx.upload_image(img)
    rng = np.random.default_rng(count)
    lo, hi = np.array([-3.0, -2.0, -1.0]), np.array([5.0, 2.0, 4.0])
    ph = np.zeros((count, 8), dtype=np.float32)
    ph[:, 3:6] = (lo + rng.random((count, 3)) * (hi - lo)).astype(np.float32)
    ph[:, 0:3] = 1.0
    m = pkg.PhotonMap(ph, lo.tolist(), hi.tolist(), 200)
    ctx.upload_photons(m.desc, m.desc, 50, False)
    pts = lo + rng.random((257, 3)) * (hi - lo)
    pos = np.ctypeslib.as_array(m.desc.photons, (count, 8))[:, 3:6].astype(np.float64)  # in the map's order
    d = pts[:, None, :] - pos[None, :, :]
    d2_all = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    for k in (1, 50):
        want = np.sort(d2_all, axis=1)[:, :min(k, count)]
        for groups in ("0", "1"):
            monkeypatch.setenv("MCRT_KNN_GROUPS", groups)
            cnt, idx, d2 = ctx.knn(0, pts, k)
            assert np.all(cnt == min(k, count))
            np.testing.assert_array_equal(d2[:, :min(k, count)], want)
            assert np.all(np.isinf(d2[:, min(k, count):])) and np.all(idx[:, min(k, count):] == 0xFFFFFFFF)
            rows = np.arange(len(pts))[:, None]
            np.testing.assert_array_equal(d2_all[rows, idx[:, :min(k, count)]], want)
    m.close()


def test_deterministic_and_shard_invariant(pkg, ctx, manifest):
    case = manifest["cases"]["hexagon_room"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])  # 192x108 @ 16 spp
    a, _ = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    b, _ = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert np.array_equal(a, b)  # same bits run to run (per-pixel sums are in sample order)
    for count, group in ((2, 8), (3, 5), (8, 8)):
        frame = np.full_like(a, np.nan)
        for i in range(count):
            c = cam.copy()
            c.shard_index, c.shard_count, c.shard_rows = i, count, group
            part = np.zeros_like(a)
            out, st = ctx.sample_image(c, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
            rows = pkg.shard_rows(c)
            frame[rows] = out[rows]
        assert np.array_equal(frame, a)  # tiling across GPUs does not change a single bit
    c = cam.copy()
    other, _ = ctx.sample_image(c, manifest["seed"] + 1, pkg.INTEGRATOR_PATH_TRACER)
    assert not np.array_equal(other, a)


def test_render_device_into_torch_buffer(pkg, ctx, manifest):
    import torch
    case = manifest["cases"]["hexagon_room_diffuse"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height = 96, 64
    host, _ = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    buf = torch.zeros((cam.height, cam.width, 3), dtype=torch.float64, device="cuda:0")
    ctx.render_device(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = ctx.render_finish()
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), host) and st["kernel_ms"] > 0


def test_c2_full_size_frame(pkg, ctx, manifest):
    """BASELINE configs[1] at its full size: 1920x1080 @ 256 spp. Checked against the reference's crop
    (rows 536-540 at full width, same per-pixel seeds) and through size-independent properties."""
    case = manifest["cases"]["hexagon_room"]
    r = [x for x in case["renders"] if x["width"] == 1920][0]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    cam = camera_for(img, r)
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    r0, r1 = r["rows"]
    _check(out[r0:r1], load_radiance(r), "C2 full-size rows %d-%d" % (r0, r1), exact=True)  # BASELINE configs[1]: the reference's bits
    assert st["paths"] == 1920 * 1080 * 256 and st["kernel_id"] == pkg.KERNEL_FLAT
    assert np.isfinite(out).all() and (out >= 0).all()
    print("C2 full frame: %.1f Mray/s, %.2f rays/path, kernel %.1f ms" %
          (st["rays"] / st["kernel_ms"] / 1e3, st["rays"] / st["paths"], st["kernel_ms"]))
    # every row range rendered as its own shard reproduces the same bits (linearity of the tiling)
    c = cam.copy()
    c.shard_index, c.shard_count, c.shard_rows = 67, 135, 8  # rows 536..543 only
    part, _ = ctx.sample_image(c, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert np.array_equal(part[536:544], out[536:544])


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_ggx", "hexagon_room_dof", "dragon_room", "veach_mis"])
def test_flat_cull_records_from_the_argument_block_or_from_lds(pkg, manifest, name):
    """The flat megakernel reads its FP32 cull records from the kernel's argument block (renderKernelFlatK, MCRT_FLAT_KARG default 1,
    scalar loads) when they fit it, from LDS otherwise or with the option at 0: the reference's bits either way."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    c = pkg.Context(0)
    try:
        c.upload_image(img)
        a, sa = c.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        c.set_option("MCRT_FLAT_KARG", 0)
        b, sb = c.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
        assert sa["kernel_id"] == sb["kernel_id"] == pkg.KERNEL_FLAT and sa["rays"] == sb["rays"]
        _check(a, load_radiance(r), "%s, cull records as a kernel argument (where they fit)" % name, exact=True)
        np.testing.assert_array_equal(a, b)
    finally:
        c.close()


def test_options_are_per_context_and_not_the_environment(pkg, manifest, monkeypatch):
    """mcrt_set_option / mcrt_get_option: the environment seeds a context's options in mcrt_create and is never read again by
    the library; two contexts in one process can run different kernel forms; NULL restores the default."""
    case = manifest["cases"]["coffee_maker_qsah"]
    img = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(img, case["renders"][0])
    monkeypatch.setenv("MCRT_KERNEL", "wf")
    a = pkg.Context(0)  # seeded with MCRT_KERNEL=wf
    monkeypatch.delenv("MCRT_KERNEL")
    b = pkg.Context(0)  # seeded without
    L = pkg.lib()
    assert a.get_option("MCRT_KERNEL") == "wf" and b.get_option("MCRT_KERNEL") is None
    assert L.mcrt_set_option(a._h, b"KERNEL", b"wf") != 0  # keys are the MCRT_* names
    st = pkg.Stats()
    frames = {}
    for name, ctx in (("a", a), ("b", b)):
        assert L.mcrt_upload_scene(ctx._h, C.byref(img.scene)) == 0  # (the raw entry points: no mirroring of os.environ by the binding)
        out = np.zeros((cam.height, cam.width, 3))
        assert L.mcrt_render(ctx._h, C.byref(cam), manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)) == 0
        frames[name] = (out, st.kernel_id)
    assert frames["a"][1] == pkg.KERNEL_WAVEFRONT and frames["b"][1] == pkg.KERNEL_LANE_SM
    np.testing.assert_array_equal(frames["a"][0], frames["b"][0])
    assert L.mcrt_set_option(a._h, b"MCRT_KERNEL", None) == 0
    out = np.zeros((cam.height, cam.width, 3))
    assert L.mcrt_render(a._h, C.byref(cam), manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)) == 0
    assert st.kernel_id == pkg.KERNEL_LANE_SM
    a.close()
    b.close()


def test_error_behaviour(pkg, manifest):
    c = pkg.Context(0)
    img = pkg.SceneImage(golden_path("hexagon_room_diffuse.mcrt"))
    cam = img.camera
    with pytest.raises(pkg.McrtError) as e:
        c.sample_image(cam, 1)
    assert "(-4)" in str(e.value)  # MCRT_ERR_NO_SCENE
    c.upload_scene(img.scene)
    with pytest.raises(pkg.McrtError) as e:
        c.sample_image(cam, 1, pkg.INTEGRATOR_PHOTON_MAPPER)
    assert "(-5)" in str(e.value)  # MCRT_ERR_NO_PHOTONS
    cam.sqrtspp = 0
    with pytest.raises(pkg.McrtError):
        c.sample_image(cam, 1)
    c.close()


@pytest.mark.parametrize("name,count,integrator", [("hexagon_room", 2, "pt"), ("coffee_maker_qsah", 3, "pt"), ("hexagon_room_pm", 2, "pm")])
def test_render_multi_equals_one_context(pkg, manifest, name, count, integrator):
    """mcrt_render_multi (one host thread per context, rows dealt over the contexts, frame assembled in host memory) with
    `count` contexts on the one GPU of the box: the frame of a single context, bit for bit."""
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    cam = camera_for(img, r)
    mode = pkg.INTEGRATOR_PHOTON_MAPPER if integrator == "pm" else pkg.INTEGRATOR_PATH_TRACER
    ctxs = [pkg.Context(0) for _ in range(count)]
    for c in ctxs:
        c.upload_image(img)
    base, st0 = ctxs[0].sample_image(cam, manifest["seed"], mode)
    out, st = pkg.render_multi(ctxs, cam, manifest["seed"], mode)
    _check(out, load_radiance(r), name + " multi")
    assert st["paths"] == st0["paths"] and st["rays"] == st0["rays"] and st["kernel_id"] == st0["kernel_id"] == expected_kernel(pkg, img, mode)
    np.testing.assert_array_equal(out, base)  # (photon-mapped frames too)
    for c in ctxs:
        c.close()
