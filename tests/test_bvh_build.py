"""Octree BVH builder: mcrt_bvh_build_octree must return the tree the reference's own builder
made ("bvh": {"type": "octree"}, the default) — the golden scene images carry the reference's LinearNode arrays and its
surface order, so rebuilding from an image's surfaces has to reproduce its node arrays bit for bit and leave the
surface order alone. Host path here; the GPU path (-m gpu) on the same scenes and on the 6.9 M-triangle C5 stand-in."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_path

OCTREE_CASES = ["hexagon_room", "hexagon_room_diffuse", "veach_mis", "metals", "oren_nayar_test", "ggx_test", "quadric"]


def _image_nodes(img):
    s = img.scene
    n = s.num_nodes
    g = lambda p, c, t: np.ctypeslib.as_array(p, shape=(c,)).astype(t, copy=True)
    return dict(bounds=g(s.node_bounds, n * 6, np.float64).reshape(n, 6), start=g(s.node_start_surface, n, np.uint32),
                count=g(s.node_num_surfaces, n, np.uint32), next=g(s.node_next_sibling, n, np.uint32))


def _check_same_tree(pkg, img, ctx=None, kind="octree", threads=0, levels=False):
    ref = _image_nodes(img)
    bvh = pkg.Bvh(img.scene, ctx=ctx, kind=kind, threads=threads, levels=levels)
    got = bvh.arrays()
    for k in ("bounds", "start", "count", "next"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(got["order"], np.arange(img.scene.num_surfaces, dtype=np.uint32))
    bvh.close()
    return len(ref["start"])


@pytest.mark.parametrize("name", OCTREE_CASES)
def test_rebuilds_the_reference_octree_bvh(pkg, manifest, name):
    img = pkg.SceneImage(golden_path(manifest["cases"][name]["image"]))
    assert _check_same_tree(pkg, img) == img.scene.num_nodes > 0


@pytest.mark.parametrize("name,kind", [("coffee_maker_qsah", "quaternary_sah"), ("coffee_maker_bsah", "binary_sah")])
@pytest.mark.parametrize("threads", [1, 0])
def test_rebuilds_the_reference_sah_bvh(pkg, manifest, name, kind, threads):
    """The binned-SAH builders restated (mcrt_bvh_build_sah): same tree as the reference's, on one thread and on all."""
    img = pkg.SceneImage(golden_path(manifest["cases"][name]["image"]))
    assert _check_same_tree(pkg, img, kind=kind, threads=threads) == img.scene.num_nodes > 0


@pytest.mark.parametrize("name,kind", [("coffee_maker_qsah", "quaternary_sah"), ("coffee_maker_bsah", "binary_sah")])
def test_level_synchronous_sah_host(pkg, manifest, name, kind):
    """mcrt_bvh_build_sah_gpu with ctx = NULL: the GPU path's level loop (mcrt_sah_shared.hpp) with its passes as host loops —
    all open nodes of a depth at once instead of the reference's recursion, same tree."""
    img = pkg.SceneImage(golden_path(manifest["cases"][name]["image"]))
    assert _check_same_tree(pkg, img, kind=kind, levels=True) == img.scene.num_nodes > 0


def test_level_synchronous_sah_equals_recursive_on_awkward_input(pkg):
    """Surfaces the split rules have trouble with: 700 triangles sharing one centroid (no usable axis, more than a leaf
    holds: arbitrarySplit, bvh.cpp:451-473), a line of centroids (the quaternary rule falls back to the binary one for the
    whole subtree, bvh.cpp:313-318) and a cloud; the level-synchronous build against the recursive restatement."""
    rng = np.random.default_rng(5)
    tris = []
    for i in range(700):  # same centroid, different sizes
        r = 0.1 + 0.001 * i
        tris.append([[-r, -r, 0.0], [r, -r, 0.0], [0.0, 2 * r, 0.0]])
    for i in range(900):  # centroids on a line along x
        x = rng.random() * 50.0
        tris.append([[x - 0.5, 1.0, 5.0], [x + 0.5, 1.0, 5.0], [x, 1.0, 5.0]])
    for i in range(3000):
        c = rng.random(3) * 20.0 - 10.0
        tris.append((c[None, :] + rng.normal(scale=0.2, size=(3, 3))).tolist())
    v = np.ascontiguousarray(np.array(tris, dtype=np.float64).reshape(-1, 9))
    n = v.shape[0]
    e = np.zeros((n, 9))
    e[:, 0:3] = v[:, 3:6] - v[:, 0:3]
    e[:, 3:6] = v[:, 6:9] - v[:, 0:3]
    kind = np.zeros(n, dtype=np.uint8)
    interp = np.zeros(n, dtype=np.uint8)
    mat = np.zeros(n, dtype=np.uint32)
    area = np.ones(n)
    m = pkg.Material()
    sc = pkg.SceneDesc()
    sc.abi_version = 2
    sc.num_surfaces = n
    sc.surf_kind = kind.ctypes.data_as(C.POINTER(C.c_uint8))
    sc.surf_interpolate = interp.ctypes.data_as(C.POINTER(C.c_uint8))
    sc.surf_material = mat.ctypes.data_as(C.POINTER(C.c_uint32))
    sc.surf_area = area.ctypes.data_as(C.POINTER(C.c_double))
    sc.surf_v = v.ctypes.data_as(C.POINTER(C.c_double))
    sc.surf_e = e.ctypes.data_as(C.POINTER(C.c_double))
    sc.num_materials = 1
    sc.materials = C.pointer(m)
    pts = v.reshape(-1, 3)
    sc.bb_min[:] = pts.min(axis=0).tolist()
    sc.bb_max[:] = pts.max(axis=0).tolist()
    for kind_name in ("quaternary_sah", "binary_sah"):
        a = pkg.Bvh(sc, kind=kind_name, threads=1)
        b = pkg.Bvh(sc, kind=kind_name, levels=True)
        ra, rb = a.arrays(), b.arrays()
        assert len(ra["start"]) > 500
        for k in ("bounds", "start", "count", "next", "order"):
            np.testing.assert_array_equal(ra[k], rb[k], err_msg="%s %s" % (kind_name, k))
        a.close()
        b.close()


@pytest.mark.parametrize("name,nodes", [("c3", 169162), ("c4", 153801), ("spaceship", 23187)])
def test_large_scene_sah(pkg, name, nodes):
    """Quaternary SAH trees of the full-size C3 / C4 stand-ins and of the spaceship cockpit, all host threads."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.image_path(name) if name != "spaceship" else os.path.join(ROOT, "oracle", "_ref", "images", "spaceship.mcrt")
    if not os.path.exists(p):
        pytest.skip("%s image not built on this machine" % name)
    img = pkg.SceneImage(p)
    t = time.perf_counter()
    assert _check_same_tree(pkg, img, kind="quaternary_sah") == nodes
    print("%s: quaternary SAH of %d surfaces rebuilt and compared in %.2f s" % (name, img.scene.num_surfaces, time.perf_counter() - t))
    t = time.perf_counter()
    assert _check_same_tree(pkg, img, kind="quaternary_sah", levels=True) == nodes
    print("%s: ... level-synchronous build on one host thread: %.2f s" % (name, time.perf_counter() - t))


def test_shuffled_surfaces_give_the_same_hits(pkg, oracle, manifest):
    """Surfaces handed over in another order: the leaves list them in that order (insertion order), nothing else
    changes — the re-ordered scene answers every ray like the original one."""
    case = manifest["cases"]["hexagon_room"]
    img = pkg.SceneImage(golden_path(case["image"]))
    s = img.scene
    n = s.num_surfaces
    perm = np.random.default_rng(3).permutation(n).astype(np.uint32)
    fake = pkg.BvhDesc()  # a "BVH" that only re-orders: builds the shuffled input scene
    fake.num_nodes, fake.num_surfaces = 0, n
    fake.order = perm.ctypes.data_as(C.POINTER(C.c_uint32))

    class _F:
        desc = fake
    shuffled = pkg.OwnedScene(s, _F)
    bvh = pkg.Bvh(shuffled.desc)
    rebuilt = bvh.apply(shuffled.desc)
    a = bvh.arrays()
    assert a["bounds"].shape[0] == s.num_nodes  # same octants
    np.testing.assert_array_equal(a["bounds"], _image_nodes(img)["bounds"])
    d = golden_path(case["kat"])
    rays = np.fromfile(os.path.join(d, "isect_rays.f64")).reshape(-1, 6)

    class _Img:  # what oracle.intersect reads
        scene = rebuilt.desc
    t, surf, uv, _ = oracle.intersect(_Img, rays[:, :3].copy(), rays[:, 3:].copy())
    t0, surf0, _, _ = oracle.intersect(img, rays[:, :3].copy(), rays[:, 3:].copy())
    np.testing.assert_array_equal(t, t0)
    hit = surf0 != 0xFFFFFFFF
    # same surface, named by its position in the re-ordered scene (exact-t ties between two surfaces aside)
    back = perm[a["order"]][surf[hit]]
    assert (back != surf0[hit]).sum() <= 5


def test_large_scene_octree(pkg):
    """The 6.9 M-triangle C5 stand-in (octree BVH built by the reference, 1 925 901 nodes), host path."""
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.image_path("c5")
    if not os.path.exists(p):
        pytest.skip("C5 image not built on this machine")
    img = pkg.SceneImage(p)
    assert _check_same_tree(pkg, img) == 1925901


@pytest.mark.gpu
@pytest.mark.parametrize("name", OCTREE_CASES)
def test_gpu_path_rebuilds_the_reference_octree_bvh(pkg, manifest, name):
    ctx = pkg.Context(0)
    img = pkg.SceneImage(golden_path(manifest["cases"][name]["image"]))
    _check_same_tree(pkg, img, ctx=ctx)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind", [("coffee_maker_qsah", "quaternary_sah"), ("coffee_maker_bsah", "binary_sah")])
def test_gpu_path_rebuilds_the_reference_sah_bvh(pkg, manifest, name, kind):
    """mcrt_bvh_build_sah_gpu: the level-synchronous binned-SAH build with its per-surface passes on the GPU."""
    img = pkg.SceneImage(golden_path(manifest["cases"][name]["image"]))
    ctx = pkg.Context(0)
    assert _check_same_tree(pkg, img, ctx=ctx, kind=kind) == img.scene.num_nodes > 0
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,nodes", [("c3", 169162), ("c4", 153801)])
def test_gpu_path_large_scene_sah(pkg, name, nodes):
    import time
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.image_path(name)
    if not os.path.exists(p) and make_large.ensure_image(name) is None:
        pytest.skip("%s image not built on this machine" % name)
    img = pkg.SceneImage(p)
    ctx = pkg.Context(0)
    pkg.Bvh(img.scene, ctx=ctx, kind="quaternary_sah").close()  # first call: allocations, module load
    assert _check_same_tree(pkg, img, ctx=ctx, kind="quaternary_sah") == nodes
    times = {}
    for label, kw in (("GPU level-synchronous", dict(ctx=ctx)), ("host, all threads, recursive", dict()), ("host, one thread, level-synchronous", dict(levels=True))):
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            pkg.Bvh(img.scene, kind="quaternary_sah", **kw).close()
            best = min(best, time.perf_counter() - t)
        times[label] = best
    print("%s: quaternary SAH of %d surfaces: %s" % (name, img.scene.num_surfaces, ", ".join("%s %.3f s" % kv for kv in times.items())))
    ctx.close()


@pytest.mark.gpu
def test_gpu_path_large_scene_and_render(pkg):
    """C5 stand-in: GPU-assisted build equals the reference's tree; timing printed."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))
    import make_large
    p = make_large.ensure_image("c5")
    if p is None:
        pytest.skip("oracle/_ref (reference binary + scene copies) not on this machine")
    img = pkg.SceneImage(p)
    ctx = pkg.Context(0)
    t = time.perf_counter()
    _check_same_tree(pkg, img, ctx=ctx)
    t_gpu = time.perf_counter() - t
    t = time.perf_counter()
    _check_same_tree(pkg, img)
    t_host = time.perf_counter() - t
    print("octree BVH of 6 898 815 triangles: GPU-assisted %.2f s, host %.2f s (incl. the comparison with the reference's arrays)" % (t_gpu, t_host))
    # the same 6.9 M surfaces under the quaternary SAH rule: level-synchronous on the GPU against the recursive host build
    t = time.perf_counter()
    a = pkg.Bvh(img.scene, ctx=ctx, kind="quaternary_sah")
    t_gpu = time.perf_counter() - t
    t = time.perf_counter()
    b = pkg.Bvh(img.scene, kind="quaternary_sah")
    t_host = time.perf_counter() - t
    ra, rb = a.arrays(), b.arrays()
    for k in ("bounds", "start", "count", "next", "order"):
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    print("quaternary SAH of the same surfaces (%d nodes): GPU level-synchronous %.2f s, host threads %.2f s" % (len(ra["start"]), t_gpu, t_host))
    a.close()
    b.close()
    ctx.close()


def test_builder_argument_errors(pkg, manifest):
    L = pkg.lib()
    img = pkg.SceneImage(golden_path(manifest["cases"]["hexagon_room"]["image"]))
    h = C.c_void_p()
    assert L.mcrt_bvh_build_sah(C.byref(img.scene), 3, 0, 0, C.byref(h)) == -1          # arity must be 2 or 4
    assert L.mcrt_bvh_build_sah(C.byref(img.scene), 4, 1, 0, C.byref(h)) == -1          # one bin cannot split
    assert L.mcrt_bvh_build_sah(None, 4, 0, 0, C.byref(h)) == -1
    assert L.mcrt_bvh_build_octree(None, None, C.byref(h)) == -1
    bad = pkg.SceneDesc()
    C.memmove(C.byref(bad), C.byref(img.scene), C.sizeof(pkg.SceneDesc))
    bad.abi_version = 1
    assert L.mcrt_bvh_build_octree(None, C.byref(bad), C.byref(h)) == -1                # descriptor of another ABI
    bvh = pkg.Bvh(img.scene)
    other = pkg.SceneImage(golden_path(manifest["cases"]["metals"]["image"]))
    s = C.c_void_p()
    assert L.mcrt_scene_with_bvh(C.byref(other.scene), C.byref(bvh.desc), C.byref(s)) == -1  # surface counts differ
    bvh.close()


def test_rebuilt_scene_renders_like_the_original(pkg, oracle, manifest):
    """mcrt_scene_with_bvh + the rebuilt tree through a whole render (oracle): the frame of the original image."""
    case = manifest["cases"]["metals"]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    from conftest import camera_for, load_radiance
    cam = camera_for(img, r)
    bvh = pkg.Bvh(img.scene, kind="quaternary_sah")  # another hierarchy over the same surfaces: same closest hits
    rebuilt = bvh.apply(img.scene)

    class _Img:
        scene = rebuilt.desc
        path = img.path

        @staticmethod
        def photons(which):
            return None

        @staticmethod
        def param(key):
            return 0
    out, _ = oracle.render(_Img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    np.testing.assert_array_equal(out, load_radiance(r))


def _same_arrays(a, b):
    A, B = a.arrays(), b.arrays()
    for k in ("bounds", "start", "count", "next", "order"):
        np.testing.assert_array_equal(A[k], B[k], err_msg=k)


@pytest.mark.parametrize("kind,bins", [("quaternary_sah", 24), ("binary_sah", 64)])
def test_level_entry_point_takes_any_bin_count(pkg, manifest, kind, bins):
    """The reference takes any "bins_per_axis" (bvh.cpp:24-40). mcrt_bvh_build_sah_gpu's level loop keeps bin tables for at most 16;
    beyond that the entry point builds through the recursive builder instead of refusing (until round 4: MCRT_ERR_UNSUPPORTED) - same
    tree as asking the recursive builder directly, and a different one from the 16-bin tree (the bins do matter)."""
    img = pkg.SceneImage(golden_path(manifest["cases"]["coffee_maker_qsah"]["image"]))
    via_levels = pkg.Bvh(img.scene, kind=kind, bins_per_axis=bins, levels=True)
    direct = pkg.Bvh(img.scene, kind=kind, bins_per_axis=bins)
    _same_arrays(via_levels, direct)
    coarse = pkg.Bvh(img.scene, kind=kind, bins_per_axis=16, levels=True)
    assert not np.array_equal(coarse.arrays()["bounds"], direct.arrays()["bounds"]) or coarse.arrays()["bounds"].shape != direct.arrays()["bounds"].shape


@pytest.mark.gpu
def test_gpu_entry_point_takes_any_bin_count(pkg, manifest):
    img = pkg.SceneImage(golden_path(manifest["cases"]["coffee_maker_qsah"]["image"]))
    ctx = pkg.Context(0)
    _same_arrays(pkg.Bvh(img.scene, ctx=ctx, kind="quaternary_sah", bins_per_axis=24), pkg.Bvh(img.scene, kind="quaternary_sah", bins_per_axis=24))
    ctx.close()
