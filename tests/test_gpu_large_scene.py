"""The out-of-LDS traversal path on a real BVH: spaceship.json (quaternary SAH built by the reference,
23 187 nodes / 68 760 triangles / 354 emissive triangles): top 512 nodes staged in LDS, everything else
read from HBM/L2. Fixture made by integration/large_scenes/make_large.py (reference output, not committed: 17 MB)."""
import os

import numpy as np
import pytest

from conftest import assert_oracle_bits, ROOT, rel_error

pytestmark = pytest.mark.gpu
IMAGES = os.path.join(ROOT, "oracle", "_ref", "images")


@pytest.fixture(scope="module")
def spaceship(pkg):
    p = os.path.join(IMAGES, "spaceship.mcrt")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/images/spaceship.mcrt not built (python __graft_entry__.py build in the build container)")
    return pkg.SceneImage(p)


def test_spaceship_matches_reference(pkg, spaceship):
    ctx = pkg.Context(0)
    ctx.upload_image(spaceship)
    cam = spaceship.camera
    cam.width, cam.height, cam.sqrtspp = 480, 270, 2
    out, st = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
    ref = np.fromfile(os.path.join(IMAGES, "spaceship.480x270_s2.f64")).reshape(270, 480, 3)
    rel = rel_error(out, ref).max(axis=2)
    print("spaceship: max rel %.3e, %.1f Mray/s, %.2f rays/path" % (rel.max(), st["rays"] / st["kernel_ms"] / 1e3, st["rays"] / st["paths"]))
    np.testing.assert_array_equal(out, ref, err_msg="spaceship: not the reference's bits")  # every operation of the path is reproduced (round 4: asin too)
    assert st["kernel_id"] == pkg.KERNEL_LANE_SM  # 23 187 nodes: the state-machine megakernel walks the tree
    ctx.close()


def test_spaceship_large_frame_goes_through_the_pipeline(pkg, spaceship):
    """Trees in memory below 65 536 nodes: the lane-state-machine megakernel for small frames, the wavefront pipeline once the call's
    rows hold 32 M path samples or more (MCRT_WF_MIN_PATHS; round 4: spaceship 1080p @ 64 spp 311 -> 228 ms). Same bits either way."""
    ctx = pkg.Context(0)
    ctx.upload_image(spaceship)
    cam = spaceship.camera
    cam.width, cam.height, cam.sqrtspp = 1920, 1080, 5   # 51.8 M path samples (MCRT_WF_MIN_PATHS: 32 M)
    full, st = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
    assert st["kernel_id"] == pkg.KERNEL_WAVEFRONT, pkg.KERNEL_NAMES.get(st["kernel_id"])
    cam.shard_rows, cam.shard_count, cam.shard_index = 8, 135, 67   # rows 536-543 alone: 0.4 M path samples
    rows = list(pkg.shard_rows(cam))
    part, st2 = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
    assert st2["kernel_id"] == pkg.KERNEL_LANE_SM, pkg.KERNEL_NAMES.get(st2["kernel_id"])
    np.testing.assert_array_equal(part[rows], full[rows])
    print("spaceship 1080p @ 25 spp: pipeline %.1f Mray/s; rows %d-%d by the megakernel: the same bits" %
          (st["rays"] / st["kernel_ms"] / 1e3, rows[0], rows[-1]))
    ctx.close()


def test_spaceship_traversal_equals_oracle(pkg, oracle, spaceship):
    rng = np.random.default_rng(7)
    s = spaceship.scene
    lo, hi = np.array(s.bb_min[:]), np.array(s.bb_max[:])
    n = 20000
    start = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ctx = pkg.Context(0)
    ctx.upload_image(spaceship)
    t, surf, uv = ctx.intersect(start, d)
    t0, s0, uv0, cnt = oracle.intersect(spaceship, start, d)
    np.testing.assert_array_equal(t, t0)
    same = surf == s0
    assert (~same).sum() <= 5  # exact-t ties only
    np.testing.assert_array_equal(uv[same], uv0[same])
    assert (surf != 0xFFFFFFFF).sum() > n // 10
    ctx.close()


# ---- BASELINE configs[2..4] at full size (stand-in meshes: integration/large_scenes/) ----

def make_large_config(name):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    return make_large.CONFIGS[name]


def _config(pkg, name):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.ensure_image(name)  # flattened here by the reference's loader + BVH builder (5-60 s)
    if p is None:
        pytest.skip("oracle/_ref (reference binary + scene copies) not on this machine")
    return pkg.SceneImage(p), make_large.CONFIGS[name], make_large.golden_path(name)


@pytest.mark.parametrize("name", ["c3", "c4", "c5", "c3:sm", "c5_s16:legacy", "c5_s16", "baroque", "lego", "pipes"])
def test_full_size_rows_match_reference(pkg, name, monkeypatch):
    """A few full-width rows of the real frame — C3: 491 592 triangles @ 1024 spp; C4: 457 200 triangles, 3840 wide
    @ 1024 spp; C5: 6 898 815 triangles, photon-mapped; baroque_table / lego_bulldozer / pipes: the reference's own scene
    files as far as their meshes exist (51 k / 123 k / 358 k triangles, up to 546 lights and 560 materials) — against the
    reference's radiance for the same rows (committed goldens made by integration/large_scenes/make_large.py)."""
    kernel = None
    if ":" in name:  # the same rows through the other kernel (default for these trees: the wavefront pipeline)
        name, kernel = name.split(":")
        monkeypatch.setenv("MCRT_KERNEL", kernel)
    img, c, golden = _config(pkg, name)
    assert (img.scene.num_surfaces, img.scene.num_nodes) == (c["surfaces"], c["nodes"])
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    cam = img.camera
    if name == "c5_s16":  # the spp bench.py --workload c5 times (256), against a reference row at that spp
        assert cam.sqrtspp == make_large_config("c5")["sqrtspp"]
        cam.sqrtspp = 16
    assert (cam.width, cam.height, cam.sqrtspp) == (c["width"], c["height"], c["sqrtspp"])
    r0, r1 = c["rows"]
    cam.shard_rows, cam.shard_count = r1 - r0, (cam.height + r1 - r0 - 1) // (r1 - r0)
    cam.shard_index = r0 // (r1 - r0)
    assert list(pkg.shard_rows(cam)) == list(range(r0, r1))
    integ = pkg.INTEGRATOR_PATH_TRACER
    if c["photon"]:
        integ = pkg.INTEGRATOR_PHOTON_MAPPER
        ctx.upload_photons(img.photons(0), img.photons(1), int(img.param("k_nearest_photons")), bool(img.param("direct_visualization")))
    out, st = ctx.sample_image(cam, 0x12345678, integ)
    ref = np.fromfile(golden).reshape(r1 - r0, c["width"], 3)
    got = out[r0:r1]  # mcrt_render writes owned rows in place
    rel = rel_error(got, ref).max(axis=2)
    differing = int((got != ref).any(axis=2).sum())
    print("%s rows %d-%d: max rel %.3e, pixels that are not the reference's bits %d / %d, %.1f Mray/s, %.2f rays/path" %
          (name, r0, r1, rel.max(), differing, rel.size, st["rays"] / st["kernel_ms"] / 1e3, st["rays"] / st["paths"]))
    if c["photon"] and kernel == "legacy":
        # The per-lane kernel keeps the reference's heap discipline and its sincosf: on the SAME map its photon-mapped rows are the
        # reference's bits (tests/test_gpu_parity.py: hexagon_room_pm, map and frame from one run of the reference). This image's
        # map was traced by another run of the reference than its golden rows - same photons, leaves filled in another, thread-
        # dependent order (tests/test_oracle_large.py) - so the judge here is the oracle, the reference's algorithm on THIS map:
        import oracle_lib
        want, _ = oracle_lib.render(img, cam, 0x12345678, integ, rows=(r0, r1))
        np.testing.assert_array_equal(got, want, err_msg="%s: the per-lane kernel's rows are not the oracle's bits" % name)
        assert rel.max() <= 1e-10
    elif c["photon"]:
        # photon-mapped rows: the k photons of an estimate are summed by a wave reduction, not in the reference's heap order
        assert rel.max() <= 1e-10
    elif differing:
        # Not the reference's bits: then the ONLY admissible reason is the closest-hit tie rule (DESIGN.md "Ties": the HIP walks return the
        # true minimum with ties to the lowest index; the reference's heap stops at `top.t >= t` and keeps the first-tested hit - they
        # part only where two surfaces are hit within an ulp, metal_bunnies' shelf against the back wall). Shown, not assumed: the
        # oracle - bit-equal to the reference on these very rows, tests/test_oracle_large.py - with THAT rule switched on
        # (oracle_set_true_minimum) must give the GPU's rows bit for bit; and the rule may only touch a few pixels, by little.
        import oracle_lib
        oracle_lib.set_true_minimum(True)
        try:
            want, _ = oracle_lib.render(img, cam, 0x12345678, integ, rows=(r0, r1))
        finally:
            oracle_lib.set_true_minimum(False)
        np.testing.assert_array_equal(got, want, err_msg="%s: neither the reference's rows nor the oracle's under the true-minimum tie rule" % name)
        assert differing <= 0.02 * rel.size and rel.max() <= 1e-4
        # (round 4, first run with every libm call restated and the exact shadow query: no frame needs this branch - C3, C4, baroque_table,
        # lego_bulldozer, pipes and the spaceship cockpit are all the reference's bits; the branch stays as the only admissible way out)
    # (the pipeline is the default for trees of 65 536 nodes or more: baroque_table and lego_bulldozer stay with the lane state machine)
    want = (pkg.KERNEL_PM_LANE if kernel == "legacy" else pkg.KERNEL_PM_WAVE) if c["photon"] else pkg.KERNEL_LANE_SM if kernel == "sm" or c["nodes"] < 65536 else pkg.KERNEL_WAVEFRONT
    assert st["kernel_id"] == want, pkg.KERNEL_NAMES.get(st["kernel_id"])
    ctx.close()


# (name, environment): the default walk on every scene; round 4's visit on the quaternary SAH tree (C3); the leaf gate wide open on
# the octree hierarchy (C5)
@pytest.mark.parametrize("name,env", [("c3", {}), ("c4", {}), ("c5", {}), ("baroque", {}), ("lego", {}), ("pipes", {}),
                                      ("c3", {"MCRT_WF_LEAN": "0"}), ("c5", {"MCRT_WF_LEAF": "1"})])
def test_full_size_traversal_equals_oracle(pkg, oracle, name, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    img, _, _ = _config(pkg, name)
    rng = np.random.default_rng(11)
    s = img.scene
    lo, hi = np.array(s.bb_min[:]), np.array(s.bb_max[:])
    n = 20000
    start = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    t, surf, uv = ctx.intersect(start, d)
    t0, s0, uv0, cnt = oracle.intersect(img, start, d)
    # C3: the shelf's back face and the back wall are coplanar (z = -1) and overlap: Möller-Trumbore gives the two
    # triangles t values one ulp apart, the second one's leaf box starts exactly at the first one's t, and the
    # reference's heap stops at `top.t >= intersect.t` (bvh.cpp:120) without looking inside. The HIP walk keeps
    # boxes with t_box == t (that is what makes exact ties order independent) and returns the true minimum.
    zfight = np.nonzero(t != t0)[0]
    assert len(zfight) <= 5
    for i in zfight:
        assert surf[i] != s0[i]
        ta = oracle.single_surface_t(img, int(surf[i]), start[i:i + 1], d[i:i + 1])[0]
        tb = oracle.single_surface_t(img, int(s0[i]), start[i:i + 1], d[i:i + 1])[0]
        assert ta == t[i] and tb == t0[i] and ta < tb and (tb - ta) <= 4 * np.spacing(tb)
    same = surf == s0
    assert (~same).sum() <= 5  # exact-t ties and the z-fight above only
    np.testing.assert_array_equal(uv[same], uv0[same])
    ctx.close()


def test_c5_emission_matches_oracle(pkg, oracle):
    """Photon emission through the 6.9 M-triangle octree BVH (emitKernel walking the quantised child blocks) against the
    oracle's emission pass: the same photons bit for bit, matched by (light, emission, bounce) key (sincos and atan2 are glibc's on
    the device, csrc/mcrt_libm.hpp: no tolerance, no unmatched keys)."""
    from conftest import sort_by_key
    img, _, _ = _config(pkg, "c5")
    want = oracle.emit_photons(img, 2000, 10.0, 0x12345678)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    got = ctx.emit_photons(2000, 10.0, 0x12345678)
    assert got["paths"] == want["paths"] == 20000
    assert got["rays"] == want["rays"]
    for name in ("global_", "caustic"):
        a, ak = sort_by_key(*got[name])
        b, bk = want[name]
        np.testing.assert_array_equal(ak, bk, err_msg="c5 %s: photon keys" % name)
        assert_oracle_bits(a, b, "c5 %s: photon records" % name)
        print("c5 %s: %d photons, identical" % (name, len(bk)))
    ctx.close()
