"""k-nearest-photon searches with k beyond 128 (LinearOctree::knnSearch takes any k, octree/linear-octree.cpp:25-117; every scene of the
reference asks for 50). Until round 4 the wave-cooperative search (csrc/mcrt_waveknn.hpp) held 256 candidates per wave and served
k <= 128; larger k fell to the per-lane legacy kernel. Its buffer now comes in two widths (4 or 16 rows of 64 candidates: k <= 128 /
k <= 768); only beyond 768 does the per-lane kernel still run. Checked here: the operator against the oracle's search on the
reference's own maps, and photon-mapped frames (flat scene; 6.9 M-triangle tree in memory; wavefront pipeline) against the oracle's
eye pass with the same k."""
import os

import numpy as np
import pytest

from conftest import golden_path, camera_for, rel_error

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture
def kernel_env():
    old = os.environ.get("MCRT_KERNEL")
    yield lambda v: os.environ.__setitem__("MCRT_KERNEL", v)
    if old is None:
        os.environ.pop("MCRT_KERNEL", None)
    else:
        os.environ["MCRT_KERNEL"] = old


@pytest.mark.parametrize("k", [129, 300, 768, 1000])
def test_gpu_knn_large_k_equals_oracle(pkg, ctx, oracle, manifest, k):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)[:1500]
        cnt, idx, d2 = ctx.knn(which, pts, k)
        ocnt, oidx, od2 = oracle.knn(img.photons(which), pts, k)
        np.testing.assert_array_equal(cnt, ocnt)
        np.testing.assert_array_equal(d2, od2)
        np.testing.assert_array_equal(idx, oidx)


@pytest.mark.parametrize("k", [200, 768])
def test_gpu_photon_mapped_frame_large_k(pkg, ctx, oracle, manifest, kernel_env, k):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    ctx.upload_photons(img.photons(0), img.photons(1), k, bool(img.param("direct_visualization")))
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 96, 54, 2
    want, _ = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER, k=k)
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st["kernel_id"] == pkg.KERNEL_PM_WAVE, pkg.KERNEL_NAMES.get(st["kernel_id"])
    rel = rel_error(out, want).max()
    print("hexagon_room_pm k = %d: max rel %.3e vs the oracle's eye pass" % (k, rel))
    assert rel <= 1e-10  # the k photons of an estimate are summed by a wave reduction, not in the reference's heap order
    kernel_env("wf")  # the pipeline's kNN launch (wfKnnKernel) with the wide buffer
    pipe, st2 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st2["kernel_id"] == pkg.KERNEL_WAVEFRONT_PM and st2["knn_searches"] == st["knn_searches"]
    np.testing.assert_array_equal(pipe, out)


def test_gpu_c5_rows_large_k(pkg, oracle):
    """The tree in memory (6.9 M triangles): the 512-lane instance with the wide buffers keeps fewer stack entries per lane in LDS."""
    from test_gpu_large_scene import _config
    img, c, _ = _config(pkg, "c5")
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    k = 300
    ctx.upload_photons(img.photons(0), img.photons(1), k, bool(img.param("direct_visualization")))
    cam = img.camera
    cam.sqrtspp = 2
    r0, r1 = 496, 498
    cam.shard_rows, cam.shard_count = r1 - r0, (cam.height + r1 - r0 - 1) // (r1 - r0)
    cam.shard_index = r0 // (r1 - r0)
    assert list(pkg.shard_rows(cam)) == list(range(r0, r1))
    out, st = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st["kernel_id"] == pkg.KERNEL_PM_WAVE, pkg.KERNEL_NAMES.get(st["kernel_id"])
    want, _ = oracle.render(img, cam, 0x12345678, pkg.INTEGRATOR_PHOTON_MAPPER, rows=(r0, r1), k=k)
    rel = rel_error(out[r0:r1], want).max()
    print("c5 rows %d-%d, k = %d: max rel %.3e, %d searches" % (r0, r1, k, rel, st["knn_searches"]))
    assert rel <= 1e-10 and st["knn_searches"] > 0
    ctx.close()


@pytest.mark.parametrize("leaf,k", [(1, 50), (2, 128), (4, 600)])
def test_gpu_knn_octree_with_tiny_leaves(pkg, ctx, manifest, leaf, k):
    """Leaves far smaller than k: more octants lie within the bound at once than the 128 frontier entries the registers of a wave
    hold (the reference's priority queue is unbounded, linear-octree.cpp:33). Until round 4: 'kNN frontier overflow'. The entries
    beyond 128 now go through a per-wave list in memory. Against a brute-force selection with the reference's distance expression."""
    img = pkg.SceneImage(golden_path(manifest["cases"]["hexagon_room_pm"]["image"]))
    ctx.upload_image(img)
    rng = np.random.default_rng(leaf * 1000 + k)
    count = 20000
    lo, hi = np.array([-3.0, -2.0, -1.0]), np.array([5.0, 2.0, 4.0])
    ph = np.zeros((count, 8), dtype=np.float32)
    ph[:, 3:6] = (lo + rng.random((count, 3)) * (hi - lo)).astype(np.float32)
    ph[:, 0:3] = 1.0
    m = pkg.PhotonMap(ph, lo.tolist(), hi.tolist(), leaf)
    ctx.upload_photons(m.desc, m.desc, k, False)
    pts = lo + rng.random((300, 3)) * (hi - lo)
    pos = np.ctypeslib.as_array(m.desc.photons, (count, 8))[:, 3:6].astype(np.float64)  # in the map's order
    d = pts[:, None, :] - pos[None, :, :]
    d2_all = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    want = np.sort(d2_all, axis=1)[:, :k]
    cnt, idx, d2 = ctx.knn(0, pts, k)
    assert np.all(cnt == k)
    np.testing.assert_array_equal(d2, want)
    rows = np.arange(len(pts))[:, None]
    np.testing.assert_array_equal(d2_all[rows, idx], want)
    m.close()


def _uniform_map(pkg, count, leaf, seed):
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-3.0, -2.0, -1.0]), np.array([5.0, 2.0, 4.0])
    ph = np.zeros((count, 8), dtype=np.float32)
    ph[:, 3:6] = (lo + rng.random((count, 3)) * (hi - lo)).astype(np.float32)
    ph[:, 0:3] = 1.0
    return pkg.PhotonMap(ph, lo.tolist(), hi.tolist(), leaf), lo, hi, rng


def _brute(m, count, pts, k):
    pos = np.ctypeslib.as_array(m.desc.photons, (count, 8))[:, 3:6].astype(np.float64)  # in the map's order
    d = pts[:, None, :] - pos[None, :, :]
    d2_all = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    return d2_all, np.sort(d2_all, axis=1)[:, :k]


@pytest.mark.parametrize("forced", [False, True])
def test_gpu_knn_frontier_beyond_the_waves_list_is_served_by_the_per_lane_kernel(pkg, ctx, manifest, forced):
    """The reference's frontier is an unbounded priority queue (linear-octree.cpp:33). A wave-cooperative search keeps 128 entries in
    registers and 1 024 in a list in memory; until round 6 a search that needed more ended the call with MCRT_ERR_UNSUPPORTED. Now the
    call is served by the per-lane kernel, whose own frontier grows on demand. The case: record lists made for k = 1 (only single
    photons are scanned whole) searched with k = 700 on leaves of one photon - thousands of octants inside the bound at once; and
    the same branch taken through the test hook, whatever the searches did. Against a brute-force selection."""
    img = pkg.SceneImage(golden_path(manifest["cases"]["hexagon_room_pm"]["image"]))
    ctx.upload_image(img)
    count, k = 60000, 700
    m, lo, hi, rng = _uniform_map(pkg, count, 1, 77)
    ctx.upload_photons(m.desc, m.desc, 1, False)
    pts = lo + rng.random((48, 3)) * (hi - lo)
    d2_all, want = _brute(m, count, pts, k)
    if forced:
        ctx.set_option("MCRT_TEST_KNN_OVERFLOW", 1)
    cnt, idx, d2 = ctx.knn(0, pts, k)
    if forced:
        ctx.set_option("MCRT_TEST_KNN_OVERFLOW", None)
    assert np.all(cnt == k)
    np.testing.assert_array_equal(d2, want)
    np.testing.assert_array_equal(d2_all[np.arange(len(pts))[:, None], idx], want)
    m.close()


def test_gpu_frame_whose_searches_overflow_is_rendered_again_by_the_per_lane_kernel(pkg, ctx, manifest, kernel_env):
    """mcrt_render_finish on a photon-mapped frame in which a wave-cooperative search ran out of frontier: the frame is rendered again by
    the per-lane kernel (until round 6: MCRT_ERR_UNSUPPORTED). No tree the tests can build fills 128 + 1 024 entries through the render
    path (its record lists are made for the k it searches with), so the overflow is raised by the test hook: the frame that comes back
    must be the per-lane kernel's, bit for bit, and say so."""
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])
    cam.width, cam.height, cam.sqrtspp = 96, 54, 1
    kernel_env("legacy")
    want, st0 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st0["kernel_id"] == pkg.KERNEL_PM_LANE
    os.environ.pop("MCRT_KERNEL")
    os.environ["MCRT_TEST_KNN_OVERFLOW"] = "1"
    try:
        out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    finally:
        os.environ.pop("MCRT_TEST_KNN_OVERFLOW")
    assert st["kernel_id"] == pkg.KERNEL_PM_LANE, pkg.KERNEL_NAMES.get(st["kernel_id"])
    np.testing.assert_array_equal(out, want)
    out2, st2 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)  # and the next frame is the fast kernel's again
    assert st2["kernel_id"] == pkg.KERNEL_PM_WAVE
    assert rel_error(out2, want).max() <= 1e-10
