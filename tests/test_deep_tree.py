"""Trees whose depth-first walk holds more pending nodes than the 128 entries the kernels' traversal stacks used to be fixed at
(the reference's frontier is an unbounded heap, bvh/bvh.cpp:80-129; until round 4 such a tree ended in MCRT_ERR_UNSUPPORTED).
The stacks are now sized per scene from the tree's own bound (HostLayout::stack_bound, mcrt_layout.hpp: the most entries ANY
depth-first walk of the tree can hold), so no ray can overflow them.

The fixture is the worst case on purpose: a BVH that is a list. N parallel triangles stacked along +x; inner node k has two
children - the leaf of the FARTHEST remaining triangle and the inner node of all nearer ones - so a ray along +x continues with
the inner child at every level and parks one leaf per level: N - 1 entries on the stack before the first primitive is tested."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_path


class ListScene:
    def __init__(self, pkg, n):
        base = pkg.SceneImage(golden_path("hexagon_room_diffuse.mcrt"))
        sc = base.scene
        self.keep = [base]

        def arr(a, ctype):
            a = np.ascontiguousarray(a)
            self.keep.append(a)
            return a.ctypes.data_as(C.POINTER(ctype))

        # triangle k (surface index k, leaf k) at x = n - k: the far ones first, so that deeper = nearer
        xs = (n - np.arange(n)).astype(np.float64)
        v = np.zeros((n, 9))
        e = np.zeros((n, 9))
        v[:, 0], v[:, 1], v[:, 2] = xs, -1.0, -1.0
        v[:, 3], v[:, 4], v[:, 5] = xs, 3.0, -1.0
        v[:, 6], v[:, 7], v[:, 8] = xs, -1.0, 3.0
        e[:, 0:3] = v[:, 3:6] - v[:, 0:3]   # E1
        e[:, 3:6] = v[:, 6:9] - v[:, 0:3]   # E2
        e[:, 6:9] = np.array([1.0, 0.0, 0.0])  # normal
        light0 = int(np.ctypeslib.as_array(sc.light_surface, (sc.num_lights,))[0])
        mats = np.ctypeslib.as_array(sc.surf_material, (sc.num_surfaces,))
        emissive = int(mats[light0])
        diffuse = int(next(m for m in mats if m != emissive))
        mat = np.full(n, diffuse, dtype=np.uint32)
        mat[0] = emissive  # the farthest triangle is the light
        # nodes in the reference's order (a node's first child follows it, the others hang on next_sibling): 2k = inner k,
        # 2k + 1 = leaf k (sibling: 2k + 2), 2k + 2 = inner k + 1; the last inner node holds the two nearest leaves
        nodes = 2 * (n - 1) + 1
        bounds = np.zeros((nodes, 6))
        start = np.zeros(nodes, dtype=np.uint32)
        count = np.zeros(nodes, dtype=np.uint32)
        nxt = np.zeros(nodes, dtype=np.uint32)

        def box(lo_x, hi_x):
            return [lo_x - 1e-6, -1.0, -1.0, hi_x + 1e-6, 3.0, 3.0]
        for k in range(n - 1):
            i = 2 * k
            bounds[i] = box(1.0, xs[k])           # inner k: triangles k .. n - 1
            bounds[i + 1] = box(xs[k], xs[k])     # leaf k
            start[i + 1], count[i + 1] = k, 1
            nxt[i + 1] = i + 2
        last = 2 * (n - 1)
        bounds[last] = box(xs[n - 1], xs[n - 1])  # where inner n - 1 would be: the nearest triangle's leaf
        start[last], count[last] = n - 1, 1
        d = pkg.SceneDesc()
        C.memmove(C.byref(d), C.byref(sc), C.sizeof(pkg.SceneDesc))
        d.num_nodes = nodes
        d.node_bounds = arr(bounds, C.c_double)
        d.node_start_surface = arr(start, C.c_uint32)
        d.node_num_surfaces = arr(count, C.c_uint32)
        d.node_next_sibling = arr(nxt, C.c_uint32)
        d.num_surfaces = n
        d.surf_kind = arr(np.zeros(n, dtype=np.uint8), C.c_uint8)
        d.surf_interpolate = arr(np.zeros(n, dtype=np.uint8), C.c_uint8)
        d.surf_material = arr(mat, C.c_uint32)
        d.surf_area = arr(np.full(n, 8.0), C.c_double)
        d.surf_v = arr(v, C.c_double)
        d.surf_e = arr(e, C.c_double)
        d.num_lights = 1
        d.light_surface = arr(np.array([0], dtype=np.uint32), C.c_uint32)
        d.light_cdf = arr(np.array([1.0]), C.c_double)
        d.bb_min[:] = [1.0, -1.0, -1.0]
        d.bb_max[:] = [float(n), 3.0, 3.0]
        self.scene = d
        self.n = n

    def photons(self, which):
        return None

    def param(self, key):
        return 0


def _rays(n, count, rng):
    """Rays along +x through the stack (every level parks a leaf), and others from random points in random directions."""
    start = np.zeros((count, 3))
    d = np.zeros((count, 3))
    start[:, 0] = -1.0
    start[:, 1:] = rng.random((count, 2)) * 0.5
    d[:] = [1.0, 0.0, 0.0]
    d[count // 2:] = rng.normal(size=(count - count // 2, 3))
    d[count // 2:, 0] = np.abs(d[count // 2:, 0]) + 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    start[3 * count // 4:, 0] = rng.random(count - 3 * count // 4) * n
    return start, d


@pytest.mark.parametrize("n", [140, 400])
def test_device_code_on_the_host_walks_a_list_shaped_tree(pkg, emu, oracle, n):
    """Host build of every walk (wave-synchronous whole-scene / top-staged, quantised blocks): the oracle's hits, no overflow."""
    s = ListScene(pkg, n)
    rng = np.random.default_rng(5)
    start, d = _rays(n, 2000, rng)
    t0, s0, uv0, _ = oracle.intersect(s, start, d)
    assert (s0[:1000] == n - 1).all()  # the axial rays end on the nearest triangle, reached last
    for stage in (0, 1, 3):
        t = np.zeros(len(start))
        surf = np.zeros(len(start), dtype=np.uint32)
        uv = np.zeros((len(start), 2))
        rc = emu.emu_intersect(C.byref(s.scene), len(start), start.ctypes.data, d.ctypes.data, stage, t.ctypes.data, surf.ctypes.data, uv.ctypes.data)
        assert rc == 0, "stage %d: rc %d (-100 = traversal stack overflow)" % (stage, rc)
        np.testing.assert_array_equal(t, t0)
        np.testing.assert_array_equal(surf, s0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [140, 400, 3000])
def test_gpu_walks_a_list_shaped_tree(pkg, oracle, n, monkeypatch):
    """On the GPU through the C ABI - mcrt_intersect (140 triangles: the whole scene staged in LDS, the wave-synchronous walk; 400 /
    3000: the tree in memory, the trace kernel) and a small frame through every integrator form. Until round 4: MCRT_ERR_UNSUPPORTED."""
    s = ListScene(pkg, n)
    rng = np.random.default_rng(5)
    start, d = _rays(n, 20000, rng)
    t0, s0, uv0, _ = oracle.intersect(s, start, d)
    ctx = pkg.Context(0)
    ctx.upload_scene(s.scene)
    t, surf, uv = ctx.intersect(start, d)
    np.testing.assert_array_equal(t, t0)
    np.testing.assert_array_equal(surf, s0)
    base = pkg.SceneImage(golden_path("hexagon_room_diffuse.mcrt"))
    cam = base.camera
    cam.eye[:] = [-2.0, 0.3, 0.3]
    cam.forward[:] = [1.0, 0.0, 0.0]
    cam.left[:] = [0.0, 0.0, -1.0]
    cam.up[:] = [0.0, 1.0, 0.0]
    cam.width, cam.height, cam.sqrtspp = 48, 32, 2
    want, _ = oracle.render(s, cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
    for kernel in (None, "sm", "wf", "legacy"):
        if kernel:
            monkeypatch.setenv("MCRT_KERNEL", kernel)
        out, st = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
        monkeypatch.delenv("MCRT_KERNEL", raising=False)
        np.testing.assert_array_equal(out, want, err_msg="kernel %s" % kernel)
    ctx.close()
