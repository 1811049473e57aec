"""ctypes wrapper of oracle/_build/liboracle.so (the plain-C restatement of the reference hot path).
CHECKER ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("paths", "rays", "node_tests", "prim_tests", "knn_searches", "knn_octants", "knn_photons", "sphere_tests")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(ROOT, "oracle", f) for f in ("mcrt_oracle.c", "mcrt_oracle.h")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"])
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.oracle_render.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                    C.c_int, vp, vp, vp, vp]
        L.oracle_intersect.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp]
        L.oracle_knn.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp]
        L.oracle_sampler.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.oracle_bsdf_kat.argtypes = [C.c_uint64, vp, vp, vp]
        L.oracle_hardware_threads.restype = C.c_int
        L.oracle_emit_photons.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, vp]
        L.oracle_image_save.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, vp, vp]
        _lib = L
    return _lib


def _ref(x):
    return C.byref(x) if x is not None else None


def render(image, cam, seed, integrator, rows=None, threads=0, per_sample=False, k=None):
    """image: SceneImage of the product binding (only its host descriptors are used).
    Returns (rgb[rows,W,3], info dict with counters and seconds)."""
    L = lib()
    r0, r1 = rows if rows else (0, cam.height)
    out = np.zeros((r1 - r0, cam.width, 3))
    samples = np.zeros((r1 - r0, cam.width, cam.sqrtspp ** 2, 3)) if per_sample else None
    cnt, sec = Counters(), C.c_double()
    g, c = image.photons(0), image.photons(1)
    rc = L.oracle_render(C.byref(image.scene), _ref(g), _ref(c), int(k or image.param("k_nearest_photons") or 50),
                         int(image.param("direct_visualization")), C.byref(cam), int(seed), int(integrator), r0, r1,
                         int(threads), out.ctypes.data, samples.ctypes.data if per_sample else None, C.byref(cnt), C.byref(sec))
    if rc != 0:
        raise RuntimeError("oracle_render failed: %d" % rc)
    info = cnt.as_dict()
    info["seconds"] = sec.value
    if per_sample:
        info["samples"] = samples
    return out, info


def set_true_minimum(on):
    """oracle_set_true_minimum: closest hits as the true minimum with ties to the lowest index (the HIP walks' rule)."""
    lib().oracle_set_true_minimum(int(bool(on)))


def intersect(image, start, direction):
    L = lib()
    start = np.ascontiguousarray(start, dtype=np.float64)
    direction = np.ascontiguousarray(direction, dtype=np.float64)
    n = start.shape[0]
    t = np.empty(n)
    surf = np.empty(n, dtype=np.uint32)
    uv = np.empty((n, 2))
    cnt = Counters()
    L.oracle_intersect(C.byref(image.scene), n, start.ctypes.data, direction.ctypes.data, t.ctypes.data, surf.ctypes.data,
                       uv.ctypes.data, C.byref(cnt))
    return t, surf, uv, cnt.as_dict()


def knn(map_desc, points, k):
    L = lib()
    points = np.ascontiguousarray(points, dtype=np.float64)
    n = points.shape[0]
    cnt = np.empty(n, dtype=np.uint32)
    idx = np.empty((n, k), dtype=np.uint32)
    d2 = np.empty((n, k))
    L.oracle_knn(C.byref(map_desc), n, points.ctypes.data, int(k), cnt.ctypes.data, idx.ctypes.data, d2.ctypes.data)
    return cnt, idx, d2


def sampler(seed, pixel, index, shuffles):
    L = lib()
    out = np.empty(7)
    L.oracle_sampler(int(seed), int(pixel), int(index), int(shuffles), out.ctypes.data)
    return out


def bsdf_kat(inputs, consts):
    L = lib()
    inputs = np.ascontiguousarray(inputs, dtype=np.float64)
    consts = np.ascontiguousarray(consts, dtype=np.float64)
    out = np.empty((inputs.shape[0], 18))
    L.oracle_bsdf_kat(inputs.shape[0], inputs.ctypes.data, consts.ctypes.data, out.ctypes.data)
    return out


def image_save(rgb, tonemapper, plain, exposure_compensation, gain_compensation):
    """Image::save: ([H,W,3] FP64) -> ([H,W,3] uint8 in B,G,R order, (exposure_factor, gain_factor))."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    bgr = np.empty((h, w, 3), dtype=np.uint8)
    factors = np.empty(2)
    rc = lib().oracle_image_save(rgb.ctypes.data, w, h, tonemapper, int(plain), exposure_compensation, gain_compensation,
                                 bgr.ctypes.data, factors.ctypes.data)
    assert rc == 0
    return bgr, (float(factors[0]), float(factors[1]))


def hardware_threads():
    return int(lib().oracle_hardware_threads())


def single_surface_t(image, surface, start, direction):
    """t at which ray(s) hit ONE surface of the image (DBL_MAX when missed): the oracle's brute-force
    loop over a one-element view of the surface arrays. Used to recognise exact-t ties."""
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    src = image.scene
    d = m.SceneDesc()
    C.memmove(C.byref(d), C.byref(src), C.sizeof(d))
    d.num_nodes = 0
    d.num_surfaces = 1

    def adv(ptr, ctype, stride):
        if not ptr:
            return ptr
        return C.cast(C.addressof(ptr.contents) + int(surface) * stride * C.sizeof(ctype), C.POINTER(ctype))

    d.surf_kind = adv(src.surf_kind, C.c_uint8, 1)
    d.surf_interpolate = adv(src.surf_interpolate, C.c_uint8, 1)
    d.surf_material = adv(src.surf_material, C.c_uint32, 1)
    d.surf_area = adv(src.surf_area, C.c_double, 1)
    d.surf_v = adv(src.surf_v, C.c_double, 9)
    d.surf_e = adv(src.surf_e, C.c_double, 9)
    d.surf_vn = adv(src.surf_vn, C.c_double, 9)

    class _One:
        scene = d

    t, _, _, _ = intersect(_One, start, direction)
    return t


def emit_photons(image, emissions, caustic_factor, seed):
    """Photon emission pass of the oracle. Returns dict(global=(photons[n,8] f32, keys[n] u64), caustic=(...),
    paths, rays), photons in (light, emission index, bounce) order."""
    L = lib()
    cap = max(1024, int(emissions * caustic_factor * 2))
    while True:
        g, gk = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
        c, ck = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
        ng, nc, paths, rays = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = L.oracle_emit_photons(C.byref(image.scene), float(emissions), float(caustic_factor), int(seed), g.ctypes.data,
                                   gk.ctypes.data, cap, C.byref(ng), c.ctypes.data, ck.ctypes.data, cap, C.byref(nc),
                                   C.byref(paths), C.byref(rays))
        if rc == 0:
            return dict(global_=(g[:ng.value], gk[:ng.value]), caustic=(c[:nc.value], ck[:nc.value]),
                        paths=paths.value, rays=rays.value)
        cap = int(max(ng.value, nc.value) * 1.1) + 16
