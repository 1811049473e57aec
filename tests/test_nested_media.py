"""Dielectrics nested deeper than the eight refraction-history entries a lane keeps in LDS (RefractionHistory, mcrt_shade.hpp;
Ray::RefractionHistory of the reference, ray/ray.cpp:74-98, is an unbounded vector): twelve concentric glass shells built from
tests/golden/ior_test.mcrt (which has four). The wavefront pipeline keeps the entries beyond 8 in rows of their own (32 entries per slot to begin with); a megakernel frame that nests deeper than
its 8 entries is rendered again through the pipeline by mcrt_render_finish (slower, correct), and a pipeline frame that nests deeper than its rows
is rendered again with four times as many (round 6: no limit but memory, like the reference's vector)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import camera_for, golden_path, rel_error


class _Scene:
    """What oracle_lib.render and Context.upload_scene need of an image: a scene descriptor (and the arrays it points into)."""

    def __init__(self, pkg, base, shells):
        sc = base.scene
        n0 = sc.num_surfaces
        kind0 = np.ctypeslib.as_array(sc.surf_kind, (n0,))
        spheres = [i for i in range(n0) if kind0[i] == 1 and sc.materials[np.ctypeslib.as_array(sc.surf_material, (n0,))[i]].transparency > 0.0]
        rest = [i for i in range(n0) if i not in spheres]
        n = shells + len(rest)
        self.keep = []

        def arr(a):
            a = np.ascontiguousarray(a)
            self.keep.append(a)
            return a

        def col(ptr, width, dtype):
            return np.ctypeslib.as_array(ptr, (n0, width) if width > 1 else (n0,)).astype(dtype)

        v0, e0 = col(sc.surf_v, 9, np.float64), col(sc.surf_e, 9, np.float64)
        area0, mat0, interp0 = col(sc.surf_area, 1, np.float64), col(sc.surf_material, 1, np.uint32), col(sc.surf_interpolate, 1, np.uint8)
        v, e = np.zeros((n, 9)), np.zeros((n, 9))
        area, mat = np.zeros(n), np.zeros(n, dtype=np.uint32)
        kind, interp = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        nm0 = sc.num_materials
        mats = (pkg.Material * (nm0 + shells))()
        for j in range(nm0):
            C.memmove(C.byref(mats[j]), C.byref(sc.materials[j]), C.sizeof(pkg.Material))
        glass = mat0[spheres[0]]
        iors = [1.4, 1.3, 1.2, 2.1, 1.5, 1.1]
        for s in range(shells):
            r = 1.0 - 0.9 * s / shells
            v[s, 0:3] = v0[spheres[0], 0:3]
            v[s, 3] = r
            area[s] = 4.0 * np.pi * r * r
            kind[s] = 1
            C.memmove(C.byref(mats[nm0 + s]), C.byref(sc.materials[glass]), C.sizeof(pkg.Material))
            mats[nm0 + s].ior = iors[s % len(iors)]
            mat[s] = nm0 + s
        remap = {}
        for k, i in enumerate(rest):
            j = shells + k
            remap[i] = j
            v[j], e[j], area[j], mat[j], kind[j], interp[j] = v0[i], e0[i], area0[i], mat0[i], kind0[i], interp0[i]
        lights = np.array([remap[int(i)] for i in np.ctypeslib.as_array(sc.light_surface, (sc.num_lights,))], dtype=np.uint32)
        cdf = np.ctypeslib.as_array(sc.light_cdf, (sc.num_lights,)).copy()
        d = pkg.SceneDesc()
        C.memmove(C.byref(d), C.byref(sc), C.sizeof(pkg.SceneDesc))
        d.num_nodes = 0
        d.num_surfaces = n
        d.surf_kind = arr(kind).ctypes.data_as(C.POINTER(C.c_uint8))
        d.surf_interpolate = arr(interp).ctypes.data_as(C.POINTER(C.c_uint8))
        d.surf_material = arr(mat).ctypes.data_as(C.POINTER(C.c_uint32))
        d.surf_area = arr(area).ctypes.data_as(C.POINTER(C.c_double))
        d.surf_v = arr(v).ctypes.data_as(C.POINTER(C.c_double))
        d.surf_e = arr(e).ctypes.data_as(C.POINTER(C.c_double))
        d.num_materials = nm0 + shells
        d.materials = C.cast(mats, C.POINTER(pkg.Material))
        d.light_surface = arr(lights).ctypes.data_as(C.POINTER(C.c_uint32))
        d.light_cdf = arr(cdf).ctypes.data_as(C.POINTER(C.c_double))
        self.keep.append(mats)
        self.scene = d
        self.base = base

    def photons(self, which):
        return None

    def param(self, key):
        return 0


class _WithBvh:
    """The same scene with a quaternary SAH tree (surfaces in the tree's order): what the lane state machine walks."""

    def __init__(self, pkg, flat):
        self.bvh = pkg.Bvh(flat.scene, kind="quaternary_sah", threads=1)
        self.owned = self.bvh.apply(flat.scene)
        self.scene = self.owned.desc
        self.flat = flat

    def photons(self, which):
        return None

    def param(self, key):
        return 0


def _setup(pkg, manifest, shells, bvh=False):
    case = manifest["cases"]["ior_test"]
    base = pkg.SceneImage(golden_path(case["image"]))
    cam = camera_for(base, case["renders"][0])
    s = _Scene(pkg, base, shells)
    return (_WithBvh(pkg, s) if bvh else s), cam


def test_oracle_sees_twelve_media(pkg, oracle, manifest):
    """The fixture does what it is for: the frame differs from the four-shell one, and paths do go through all the shells."""
    s12, cam = _setup(pkg, manifest, 12)
    s4, _ = _setup(pkg, manifest, 4)
    a, _ = oracle.render(s12, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    b, _ = oracle.render(s4, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert np.isfinite(a).all() and rel_error(a, b).max() > 1e-3


def test_wavefront_device_code_on_the_host_keeps_deep_histories(pkg, emu, oracle, manifest):
    s12, cam = _setup(pkg, manifest, 12)
    ref, _ = oracle.render(s12, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    out = np.zeros((cam.height, cam.width, 3))
    cnt = (C.c_uint64 * 8)()
    assert emu.emu_render_wf(C.byref(s12.scene), C.byref(cam), manifest["seed"], 500, cam.height, out.ctypes.data, cnt) == 0
    np.testing.assert_array_equal(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [None, "wf", "sm"])
def test_gpu_keeps_deep_histories(pkg, oracle, manifest, kernel):
    s12, cam = _setup(pkg, manifest, 12, bvh=(kernel == "sm"))
    ref, _ = oracle.render(s12, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    ctx = pkg.Context(0)
    old = os.environ.pop("MCRT_KERNEL", None)
    try:
        if kernel:
            os.environ["MCRT_KERNEL"] = kernel
        ctx.upload_scene(s12.scene)
        out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    finally:
        os.environ.pop("MCRT_KERNEL", None)
        if old is not None:
            os.environ["MCRT_KERNEL"] = old
    assert st["kernel_id"] == 4  # whatever was asked for, the frame that holds comes from the pipeline
    # (the scene has a sky; since round 4 the device's asin is glibc's too, csrc/mcrt_libm.hpp: the oracle's bits)
    np.testing.assert_array_equal(out, ref)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shells", [40, 150])
def test_gpu_keeps_histories_beyond_the_first_rows(pkg, oracle, manifest, shells):
    """Forty and a hundred and fifty nested shells: deeper than the 32 entries a pipeline slot starts with. Until round 6 such a frame
    ended with MCRT_ERR_UNSUPPORTED; now mcrt_render_finish renders it again with four times the deep rows (32 -> 128 -> 512) until
    the histories fit - the reference's vector is unbounded (ray/ray.cpp:74-98) - and the frame is the oracle's bits."""
    s, cam = _setup(pkg, manifest, shells)
    ref, _ = oracle.render(s, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    ctx = pkg.Context(0)
    ctx.upload_scene(s.scene)
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert st["kernel_id"] == 4
    np.testing.assert_array_equal(out, ref)
    out2, _ = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)  # (the rows stay with the context)
    np.testing.assert_array_equal(out2, ref)
    ctx.close()


@pytest.mark.gpu
def test_gpu_emission_reports_histories_beyond_its_limit(pkg, oracle, manifest):
    """The photon emission pass keeps 8 refraction-history entries per lane: photon paths through the twelve shells nest deeper.
    Until round 4 the kernel dropped the overflow (a photon then carried a wrong external IOR, silently); now the call fails. With
    four shells - the reference's own ior_test - the lists are the oracle's."""
    s12, _ = _setup(pkg, manifest, 12)
    ctx = pkg.Context(0)
    ctx.upload_scene(s12.scene)
    with pytest.raises(pkg.McrtError) as e:
        ctx.emit_photons(500000, 10.0, manifest["seed"])  # (the shells fill 0.08 % of the lamp's sphere of directions: a few thousand of the 5 M paths enter them)
    assert "nested dielectric" in str(e.value)
    ctx.close()
    from conftest import sort_by_key
    s4, _ = _setup(pkg, manifest, 4)
    want = oracle.emit_photons(s4, 2000, 10.0, manifest["seed"])
    ctx = pkg.Context(0)
    ctx.upload_scene(s4.scene)
    got = ctx.emit_photons(2000, 10.0, manifest["seed"])
    for name in ("global_", "caustic"):
        a, ak = sort_by_key(*got[name])
        b, bk = want[name]
        np.testing.assert_array_equal(ak, bk)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    ctx.close()
