"""Host-side Python binding of libmcrt_hip.so (the C ABI in include/mcrt.h).

The reference (linusmossberg/monte-carlo-ray-tracer) is a C++ program whose render seam is
``Camera::sampleImage()`` (source/camera/camera.cpp:101-145); the product is the HIP library behind
the C ABI, and the C++ host driver lives in ``host/``. This module is the thin ctypes layer the
tests and bench.py use to reach the same entry points; it contains no rendering logic and no CPU
fallback: if the shared library (built in-tree by ``build.py`` / ``__graft_entry__.build()``) is
missing, importing :func:`lib` raises.

Import with ``importlib.import_module("monte-carlo-ray-tracer_amd")`` (the directory name is not a
Python identifier).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCRT_TOLERANCE_BUILD=1 in the environment at import: the opt-in tolerance library (build.py LIB_TOL: FP64 contraction + the platform's
# libm; frames within BASELINE.json's 1e-4 relative bar instead of the reference's bits). Default: the exact library.
TOLERANCE_BUILD = os.environ.get("MCRT_TOLERANCE_BUILD") == "1"
LIB_PATH = os.path.join(_HERE, "csrc", "libmcrt_hip_tol.so" if TOLERANCE_BUILD else "libmcrt_hip.so")
ABI_VERSION = 2

FILM_FILTERS = {"box": 0, "mitchell-netravali": 1, "catmull-rom": 2, "b-spline": 3, "hermite": 4, "gaussian": 5, "lanczos": 6}
INTEGRATOR_PATH_TRACER = 0
INTEGRATOR_PHOTON_MAPPER = 1
LIBM_SINCOS, LIBM_SIN, LIBM_COS, LIBM_ASIN, LIBM_ATAN2, LIBM_SINCOSF, LIBM_POW = range(7)  # mcrt_libm function selectors (include/mcrt.h MCRT_LIBM_*)
# mcrt_stats.kernel_id (include/mcrt.h MCRT_KERNEL_*)
KERNEL_NONE, KERNEL_FLAT, KERNEL_WAVESYNC, KERNEL_LANE_SM, KERNEL_WAVEFRONT, KERNEL_PM_WAVE, KERNEL_PM_LANE, KERNEL_WAVEFRONT_PM = range(8)
KERNEL_NAMES = {KERNEL_NONE: "none", KERNEL_FLAT: "renderKernelFlatK (flat loop; renderKernel<path_tracer, flat> when the cull records do not fit the argument block)", KERNEL_WAVESYNC: "renderKernel<path_tracer>",
                KERNEL_LANE_SM: "renderKernelSM", KERNEL_WAVEFRONT: "wfTraceKernel + wfShadeKernel", KERNEL_PM_WAVE: "renderKernelPM",
                KERNEL_PM_LANE: "renderKernel<photon_mapper>", KERNEL_WAVEFRONT_PM: "wfTraceKernel + wfKnnKernel + wfShadeKernel"}
SURF_TRIANGLE, SURF_SPHERE = 0, 1
NO_SURFACE = 0xFFFFFFFF

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_fp = C.POINTER(C.c_float)


class Material(C.Structure):
    """mcrt_material — one record per reference Material (material/material.hpp:9-55)."""
    _fields_ = [
        ("reflectance", C.c_double * 3), ("specular_reflectance", C.c_double * 3),
        ("transmittance", C.c_double * 3), ("emittance", C.c_double * 3),
        ("roughness", C.c_double), ("specular_roughness", C.c_double), ("ior", C.c_double),
        ("transparency", C.c_double), ("A", C.c_double), ("B", C.c_double), ("a", C.c_double * 2),
        ("ior_real", C.c_double * 3), ("ior_imag", C.c_double * 3),
        ("flags", C.c_uint32), ("reserved", C.c_uint32),
    ]


class SceneDesc(C.Structure):
    """mcrt_scene_desc."""
    _fields_ = [
        ("abi_version", C.c_uint32), ("num_nodes", C.c_uint32),
        ("node_bounds", _dp), ("node_start_surface", _u32p), ("node_num_surfaces", _u32p),
        ("node_next_sibling", _u32p),
        ("num_surfaces", C.c_uint32),
        ("surf_kind", _u8p), ("surf_interpolate", _u8p), ("surf_material", _u32p),
        ("surf_area", _dp), ("surf_v", _dp), ("surf_e", _dp), ("surf_vn", _dp),
        ("num_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("num_lights", C.c_uint32), ("light_surface", _u32p), ("light_cdf", _dp),
        ("scene_ior", C.c_double), ("bb_min", C.c_double * 3), ("bb_max", C.c_double * 3),
        ("num_quadrics", C.c_uint32), ("quadrics", _dp),
    ]


class BvhDesc(C.Structure):
    """mcrt_bvh_desc."""
    _fields_ = [
        ("num_nodes", C.c_uint32), ("node_bounds", _dp), ("node_start_surface", _u32p), ("node_num_surfaces", _u32p),
        ("node_next_sibling", _u32p), ("num_surfaces", C.c_uint32), ("order", _u32p),
    ]


class PhotonMapDesc(C.Structure):
    """mcrt_photon_map_desc."""
    _fields_ = [
        ("num_octants", C.c_uint32), ("octant_bounds", _dp), ("octant_start_data", _u64p),
        ("octant_contained_data", _u64p), ("octant_next_sibling", _u32p), ("octant_leaf", _u8p),
        ("num_photons", C.c_uint64), ("photons", _fp),
    ]


class CameraDesc(C.Structure):
    """mcrt_camera_desc — the Camera fields read by samplePixel (camera/camera.cpp:66-99)."""
    _fields_ = [
        ("eye", C.c_double * 3), ("forward", C.c_double * 3), ("left", C.c_double * 3),
        ("up", C.c_double * 3),
        ("focal_length", C.c_double), ("sensor_width", C.c_double),
        ("aperture_radius", C.c_double), ("focus_distance", C.c_double),
        ("thin_lens", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("sqrtspp", C.c_uint32),
        ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("shard_rows", C.c_uint32),
        ("film_filter", C.c_uint32), ("film_radius", C.c_double), ("film_cache_size", C.c_uint32), ("reserved", C.c_uint32),
    ]

    def copy(self):
        c = CameraDesc()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(CameraDesc))
        return c


class Stats(C.Structure):
    """mcrt_stats."""
    _fields_ = [
        ("paths", C.c_uint64), ("rays", C.c_uint64), ("node_tests", C.c_uint64),
        ("prim_tests", C.c_uint64), ("knn_searches", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
        ("kernel_launches", C.c_uint32), ("kernel_id", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


TONEMAP_HABLE, TONEMAP_ACES = 0, 1
TONEMAPPERS = {"HABLE": TONEMAP_HABLE, "ACES": TONEMAP_ACES}


class ImageDesc(C.Structure):
    """mcrt_image_desc — the camera's "image" object (camera/image.cpp:10-35)."""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("tonemapper", C.c_uint32), ("plain", C.c_uint32),
        ("exposure_compensation", C.c_double), ("gain_compensation", C.c_double),
    ]

    @classmethod
    def make(cls, width, height, tonemapper="HABLE", plain=False, exposure_compensation=0.0, gain_compensation=0.0):
        # image.cpp:25-34: the name is upper-cased; "ACES" selects filmicACES, anything else filmicHable
        tm = tonemapper if isinstance(tonemapper, int) else TONEMAPPERS.get(str(tonemapper).upper(), TONEMAP_HABLE)
        return cls(int(width), int(height), tm, int(bool(plain)), float(exposure_compensation), float(gain_compensation))


class PhotonEmission(C.Structure):
    """mcrt_photon_emission."""
    _fields_ = [
        ("global_count", C.c_uint64), ("caustic_count", C.c_uint64),
        ("global_photons", _fp), ("caustic_photons", _fp),
        ("global_keys", _u64p), ("caustic_keys", _u64p),
        ("emission_paths", C.c_uint64), ("rays", C.c_uint64), ("kernel_ms", C.c_double),
    ]


class PhotonPassStats(C.Structure):
    _fields_ = [("global_count", C.c_uint64), ("caustic_count", C.c_uint64), ("global_octants", C.c_uint64), ("caustic_octants", C.c_uint64),
                ("emission_paths", C.c_uint64), ("rays", C.c_uint64), ("emission_ms", C.c_double), ("sort_ms", C.c_double),
                ("octant_ms", C.c_double), ("finish_ms", C.c_double), ("total_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PhotonEmissionDevice(C.Structure):
    _fields_ = [("global_count", C.c_uint64), ("caustic_count", C.c_uint64), ("d_global_photons", C.c_void_p), ("d_caustic_photons", C.c_void_p),
                ("emission_paths", C.c_uint64), ("rays", C.c_uint64), ("kernel_ms", C.c_double)]


def render_multi(contexts, cam, global_seed, integrator=INTEGRATOR_PATH_TRACER):
    """mcrt_render_multi: one frame over several contexts (one per GPU, scene already uploaded), one host thread each ->
    (image[H,W,3] float64, stats dict)."""
    out = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
    st = Stats()
    handles = (C.c_void_p * len(contexts))(*[c._h.value for c in contexts])
    rc = lib().mcrt_render_multi(handles, len(contexts), C.byref(cam), int(global_seed), int(integrator), _ptr(out, C.c_double), C.byref(st))
    contexts[0]._check(rc, "mcrt_render_multi")
    return out, st.as_dict()


def photon_pass_multi(contexts, emissions, caustic_factor, global_seed, bb_min, bb_max, max_photons_per_leaf=200, k_nearest=50,
                      direct_visualization=False):
    """mcrt_photon_pass_multi: the photon pass sharded over several contexts of this process (emission shards exchanged on device
    pointers, the same maps built in every context). Returns one stats dict per context."""
    for c in contexts:
        c._sync_env()
    st = (PhotonPassStats * len(contexts))()
    lo, hi = (C.c_double * 3)(*bb_min), (C.c_double * 3)(*bb_max)
    handles = (C.c_void_p * len(contexts))(*[c._h.value for c in contexts])
    rc = lib().mcrt_photon_pass_multi(handles, len(contexts), float(emissions), float(caustic_factor), int(global_seed), lo, hi,
                                      int(max_photons_per_leaf), int(k_nearest), 1 if direct_visualization else 0, st)
    contexts[0]._check(rc, "mcrt_photon_pass_multi")
    return [s.as_dict() for s in st]


def tga_save(path, bgr):
    """mcrt_tga_save: the reference's .tga (HeaderTGA + B,G,R bytes) for a [H,W,3] uint8 array."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    rc = lib().mcrt_tga_save(os.fsencode(path), bgr.shape[1], bgr.shape[0], bgr.ctypes.data)
    if rc != 0:
        raise McrtError("mcrt_tga_save(%s) failed: %d" % (path, rc))


class McrtError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmcrt_hip.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McrtError(
            "%s is missing: build it with `python __graft_entry__.py build` "
            "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.mcrt_abi_version.restype = C.c_uint32
    L.mcrt_create.argtypes = [C.POINTER(vp), C.c_int]
    if hasattr(L, "mcrt_device_count"):  # (absent from libraries older than round 5: tools/ab_builds.sh loads those too)
        L.mcrt_device_count.argtypes = []
        L.mcrt_device_count.restype = C.c_int
    L.mcrt_destroy.argtypes = [vp]
    L.mcrt_destroy.restype = None
    L.mcrt_last_error.argtypes = [vp]
    L.mcrt_last_error.restype = C.c_char_p
    L.mcrt_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.mcrt_get_option.argtypes = [vp, C.c_char_p]
    L.mcrt_get_option.restype = C.c_char_p
    L.mcrt_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.mcrt_upload_photons.argtypes = [vp, C.POINTER(PhotonMapDesc), C.POINTER(PhotonMapDesc),
                                      C.c_uint32, C.c_int]
    L.mcrt_render.argtypes = [vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, _dp, C.POINTER(Stats)]
    L.mcrt_render_device.argtypes = [vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, vp, vp]
    L.mcrt_render_finish.argtypes = [vp, C.POINTER(Stats)]
    L.mcrt_shard_rows.argtypes = [C.POINTER(CameraDesc), _u32p]
    L.mcrt_shard_rows.restype = C.c_uint32
    L.mcrt_emit_photons.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.POINTER(PhotonEmission)]
    L.mcrt_emit_photons_shard.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PhotonEmission)]
    L.mcrt_photon_pass_device.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, _dp, _dp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(PhotonPassStats)]
    L.mcrt_emit_photons_device.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PhotonEmissionDevice)]
    L.mcrt_upload_photons_device.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, _dp, _dp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(PhotonPassStats)]
    L.mcrt_photon_map_download.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.mcrt_intersect.argtypes = [vp, C.c_uint64, _dp, _dp, _dp, _u32p, _dp]
    L.mcrt_sampler.argtypes = [vp, C.c_uint64, _u32p, _u32p, C.c_uint32, C.c_uint32, _dp]
    L.mcrt_knn.argtypes = [vp, C.c_int, C.c_uint64, _dp, C.c_uint32, _u32p, _u32p, _dp]
    L.mcrt_bsdf.argtypes = [vp, C.c_uint64, _dp, _dp, _dp]
    L.mcrt_libm.argtypes = [vp, C.c_int, C.c_uint64, _dp, _dp, _dp, _dp]
    L.mcrt_bvh_build_octree.argtypes = [vp, C.POINTER(SceneDesc), C.POINTER(vp)]
    L.mcrt_bvh_build_sah.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.mcrt_bvh_build_sah_gpu.argtypes = [vp, C.POINTER(SceneDesc), C.c_int, C.c_uint32, C.POINTER(vp)]
    L.mcrt_bvh_get.argtypes = [vp]
    L.mcrt_bvh_get.restype = C.POINTER(BvhDesc)
    L.mcrt_bvh_free.argtypes = [vp]
    L.mcrt_scene_with_bvh.argtypes = [C.POINTER(SceneDesc), C.POINTER(BvhDesc), C.POINTER(vp)]
    L.mcrt_scene_get.argtypes = [vp]
    L.mcrt_scene_get.restype = C.POINTER(SceneDesc)
    L.mcrt_scene_free.argtypes = [vp]
    L.mcrt_photon_map_build.argtypes = [_fp, C.c_uint64, _dp, _dp, C.c_uint32, C.POINTER(vp)]
    L.mcrt_photon_map_build_gpu.argtypes = [vp, _fp, C.c_uint64, _dp, _dp, C.c_uint32, C.POINTER(vp)]
    L.mcrt_photon_map_get.argtypes = [vp]
    L.mcrt_photon_map_get.restype = C.POINTER(PhotonMapDesc)
    L.mcrt_photon_map_free.argtypes = [vp]
    L.mcrt_photon_map_free.restype = None
    L.mcrt_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(CameraDesc), C.c_uint32, C.c_int, _dp, C.POINTER(Stats)]
    if hasattr(L, "mcrt_photon_pass_multi"):  # (round 6; absent from older libraries that tools/ab_builds.sh swaps in)
        L.mcrt_photon_pass_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.c_double, C.c_double, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.c_uint32, C.c_uint32, C.c_int, C.POINTER(PhotonPassStats)]
    L.mcrt_render_film_device.argtypes = [vp, C.POINTER(CameraDesc), C.c_uint32, C.c_int, vp, vp]
    L.mcrt_film_resolve_device.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.mcrt_tonemap_device.argtypes = [vp, vp, C.POINTER(ImageDesc), vp, _dp, vp]
    L.mcrt_tonemap.argtypes = [vp, _dp, C.POINTER(ImageDesc), vp, _dp]
    L.mcrt_tga_save.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.mcrt_image_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.mcrt_image_free.argtypes = [vp]
    L.mcrt_image_free.restype = None
    L.mcrt_image_scene.argtypes = [vp]
    L.mcrt_image_scene.restype = C.POINTER(SceneDesc)
    L.mcrt_image_camera.argtypes = [vp]
    L.mcrt_image_camera.restype = C.POINTER(CameraDesc)
    L.mcrt_image_photons.argtypes = [vp, C.c_int]
    L.mcrt_image_photons.restype = C.POINTER(PhotonMapDesc)
    L.mcrt_image_param.argtypes = [vp, C.c_char_p]
    L.mcrt_image_param.restype = C.c_uint64
    if L.mcrt_abi_version() != ABI_VERSION:
        raise McrtError("libmcrt_hip.so ABI %d != binding ABI %d" % (L.mcrt_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class SceneImage:
    """A scene image (*.mcrt) loaded through mcrt_image_load: flattened Scene/BVH/Camera (+ photon
    maps) as written by the flattener inside the reference host."""

    def __init__(self, path):
        self._lib = lib()
        self._h = C.c_void_p()
        rc = self._lib.mcrt_image_load(os.fsencode(path), C.byref(self._h))
        if rc != 0:
            raise McrtError("mcrt_image_load(%s) failed: %d" % (path, rc))
        self.path = path

    @property
    def scene(self):
        return self._lib.mcrt_image_scene(self._h).contents

    @property
    def camera(self):
        return self._lib.mcrt_image_camera(self._h).contents.copy()

    def photons(self, which):
        p = self._lib.mcrt_image_photons(self._h, which)
        return p.contents if p else None

    def param(self, key):
        return int(self._lib.mcrt_image_param(self._h, key.encode()))

    def close(self):
        if self._h:
            self._lib.mcrt_image_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bvh:
    """mcrt_bvh: the reference's default ("octree") BVH of a scene's surfaces, built by sorting centroid path codes
    (mcrt_bvh_build_octree; with a Context the per-surface work and the sort run on its GPU)."""

    def __init__(self, scene_desc, ctx=None, kind="octree", bins_per_axis=0, threads=0, levels=False):
        self._lib = lib()
        self._h = C.c_void_p()
        bins = int(bins_per_axis) if bins_per_axis else (8 if kind == "quaternary_sah" else 16)
        if kind in ("binary_sah", "quaternary_sah") and (ctx is not None or levels):
            # level-synchronous build: on the GPU of ctx, or (levels=True, no ctx) the same passes as host loops. Its per-node bin tables
            # hold at most 16 bins per axis (the reference's defaults are 16 and 8); for more the C entry point itself hands the scene to
            # the recursive host builder, which has no limit and builds the same tree.
            rc = self._lib.mcrt_bvh_build_sah_gpu(ctx._h if ctx is not None else None, C.byref(scene_desc), 2 if kind == "binary_sah" else 4,
                                                  int(bins_per_axis), C.byref(self._h))
            if rc != 0:
                if ctx is not None:
                    ctx._check(rc, "mcrt_bvh_build_sah_gpu")
                raise McrtError("mcrt_bvh_build_sah_gpu failed: %d" % rc)
            return
        if kind in ("binary_sah", "quaternary_sah"):  # the reference's binned-SAH builders on all host threads
            rc = self._lib.mcrt_bvh_build_sah(C.byref(scene_desc), 2 if kind == "binary_sah" else 4, int(bins_per_axis), int(threads), C.byref(self._h))
            if rc != 0:
                raise McrtError("mcrt_bvh_build_sah failed: %d" % rc)
            return
        rc = self._lib.mcrt_bvh_build_octree(ctx._h if ctx is not None else None, C.byref(scene_desc), C.byref(self._h))
        if rc != 0:
            if ctx is not None:
                ctx._check(rc, "mcrt_bvh_build_octree")
            raise McrtError("mcrt_bvh_build_octree failed: %d" % rc)

    @property
    def desc(self):
        return self._lib.mcrt_bvh_get(self._h).contents

    def arrays(self):
        d = self.desc
        n = d.num_nodes

        def grab(ptr, count, dtype):
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True) if count else np.zeros(0, dtype)
        return dict(bounds=grab(d.node_bounds, n * 6, np.float64).reshape(n, 6), start=grab(d.node_start_surface, n, np.uint32),
                    count=grab(d.node_num_surfaces, n, np.uint32), next=grab(d.node_next_sibling, n, np.uint32),
                    order=grab(d.order, d.num_surfaces, np.uint32))

    def apply(self, scene_desc):
        """mcrt_scene_with_bvh -> OwnedScene (surfaces in BVH order, lights re-indexed, this BVH's nodes)."""
        return OwnedScene(scene_desc, self)

    def close(self):
        if self._h:
            self._lib.mcrt_bvh_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OwnedScene:
    def __init__(self, scene_desc, bvh):
        self._lib = lib()
        self._h = C.c_void_p()
        self._keep = (scene_desc, bvh)  # the copy still points into their material / light / quadric / node arrays
        rc = self._lib.mcrt_scene_with_bvh(C.byref(scene_desc), C.byref(bvh.desc), C.byref(self._h))
        if rc != 0:
            raise McrtError("mcrt_scene_with_bvh failed: %d" % rc)

    @property
    def desc(self):
        return self._lib.mcrt_scene_get(self._h).contents

    def close(self):
        if self._h:
            self._lib.mcrt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PhotonMap:
    """mcrt_photon_map: linear photon octree built from a photon list — on the host (mcrt_photon_map_build) or,
    given a Context, with its GPU (mcrt_photon_map_build_gpu: cell codes, radix sort, gather, leaf boxes)."""

    @classmethod
    def _from_handle(cls, handle):
        m = cls.__new__(cls)
        m._lib = lib()
        m._h = handle
        return m

    def __init__(self, photons, bb_min, bb_max, max_photons_per_leaf=200, ctx=None):
        self._lib = lib()
        self._h = C.c_void_p()
        ph = np.ascontiguousarray(photons, dtype=np.float32).reshape(-1, 8)
        lo = (C.c_double * 3)(*bb_min)
        hi = (C.c_double * 3)(*bb_max)
        if ctx is None:
            rc = self._lib.mcrt_photon_map_build(_ptr(ph, C.c_float), ph.shape[0], lo, hi, int(max_photons_per_leaf), C.byref(self._h))
            if rc != 0:
                raise McrtError("mcrt_photon_map_build failed: %d" % rc)
        else:
            ctx._check(self._lib.mcrt_photon_map_build_gpu(ctx._h, _ptr(ph, C.c_float), ph.shape[0], lo, hi, int(max_photons_per_leaf),
                                                           C.byref(self._h)), "mcrt_photon_map_build_gpu")

    def arrays(self):
        """The descriptor as numpy copies: dict(bounds[n,6], start[n], contained[n], next[n], leaf[n], photons[m,8])."""
        d = self.desc
        n, m = d.num_octants, d.num_photons

        def grab(ptr, count, dtype):
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True) if count else np.zeros(0, dtype)
        return dict(bounds=grab(d.octant_bounds, n * 6, np.float64).reshape(n, 6), start=grab(d.octant_start_data, n, np.uint64),
                    contained=grab(d.octant_contained_data, n, np.uint64), next=grab(d.octant_next_sibling, n, np.uint32),
                    leaf=grab(d.octant_leaf, n, np.uint8), photons=grab(d.photons, m * 8, np.float32).reshape(m, 8))

    @property
    def desc(self):
        return self._lib.mcrt_photon_map_get(self._h).contents

    def close(self):
        if self._h:
            self._lib.mcrt_photon_map_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """mcrt_ctx: one per GPU / process rank. Mirrors the reference's render seam:
    ``Context.sample_image(camera)`` stands where ``Camera::sampleImage()`` stood."""

    def __init__(self, device_id=0):
        self._lib = lib()
        self._h = C.c_void_p()
        rc = self._lib.mcrt_create(C.byref(self._h), int(device_id))
        if rc != 0:
            msg = self._lib.mcrt_last_error(None)
            raise McrtError("mcrt_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))

        self._env = {k: v for k, v in os.environ.items() if k.startswith("MCRT_")}  # what mcrt_create seeded the options with

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.mcrt_last_error(self._h)
            raise McrtError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def set_option(self, key, value):
        """mcrt_set_option: a run-time option of this context (value None = back to the default)."""
        self._check(self._lib.mcrt_set_option(self._h, key.encode(), None if value is None else str(value).encode()), "mcrt_set_option")

    def get_option(self, key):
        v = self._lib.mcrt_get_option(self._h, key.encode())
        return v.decode() if v is not None else None

    def _sync_env(self):
        """The library reads the MCRT_* environment once, in mcrt_create. The tests and A/B tools of this repo switch kernels by
        changing os.environ between calls; the binding mirrors such changes into mcrt_set_option before a call that launches."""
        now = {k: v for k, v in os.environ.items() if k.startswith("MCRT_")}
        if now != self._env:
            for k in set(self._env) - set(now):
                self.set_option(k, None)
            for k, v in now.items():
                if self._env.get(k) != v:
                    self.set_option(k, v)
            self._env = now

    def upload_scene(self, scene_desc):
        self._sync_env()
        self._check(self._lib.mcrt_upload_scene(self._h, C.byref(scene_desc)), "mcrt_upload_scene")

    def upload_photons(self, global_map, caustic_map, k_nearest, direct_visualization=False):
        self._sync_env()
        g = C.byref(global_map) if global_map is not None else None
        c = C.byref(caustic_map) if caustic_map is not None else None
        self._check(self._lib.mcrt_upload_photons(self._h, g, c, int(k_nearest),
                                                  int(bool(direct_visualization))),
                    "mcrt_upload_photons")

    def upload_image(self, image):
        """Upload everything a SceneImage holds (scene + photon maps if present)."""
        self.upload_scene(image.scene)
        g, c = image.photons(0), image.photons(1)
        if g is not None or c is not None:
            self.upload_photons(g, c, image.param("k_nearest_photons") or 50,
                                bool(image.param("direct_visualization")))

    def sample_image(self, cam, global_seed, integrator=INTEGRATOR_PATH_TRACER):
        """mcrt_render -> (image[H,W,3] float64, stats dict)."""
        self._sync_env()
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
        st = Stats()
        self._check(self._lib.mcrt_render(self._h, C.byref(cam), int(global_seed), int(integrator),
                                          _ptr(out, C.c_double), C.byref(st)), "mcrt_render")
        return out, st.as_dict()

    def render_device(self, cam, global_seed, integrator, device_ptr, stream=None):
        self._sync_env()
        self._check(self._lib.mcrt_render_device(self._h, C.byref(cam), int(global_seed),
                                                 int(integrator), C.c_void_p(int(device_ptr)),
                                                 C.c_void_p(int(stream)) if stream else None),
                    "mcrt_render_device")

    def render_film_device(self, cam, global_seed, integrator, rgbw_ptr, stream=None):
        """mcrt_render_film_device: this shard's splats into a full-frame RGBW device buffer (width*height*4 doubles)."""
        self._sync_env()
        self._check(self._lib.mcrt_render_film_device(self._h, C.byref(cam), int(global_seed), int(integrator), C.c_void_p(int(rgbw_ptr)),
                                                      C.c_void_p(int(stream)) if stream else None), "mcrt_render_film_device")

    def film_resolve_device(self, width, height, rgbw_ptr, out_ptr, stream=None):
        """mcrt_film_resolve_device: Splat::get over the (summed) RGBW buffer -> width*height*3 doubles."""
        self._check(self._lib.mcrt_film_resolve_device(self._h, int(width), int(height), C.c_void_p(int(rgbw_ptr)), C.c_void_p(int(out_ptr)),
                                                       C.c_void_p(int(stream)) if stream else None), "mcrt_film_resolve_device")

    def render_finish(self):
        st = Stats()
        self._check(self._lib.mcrt_render_finish(self._h, C.byref(st)), "mcrt_render_finish")
        return st.as_dict()

    def tonemap(self, rgb, image):
        """mcrt_tonemap: Image::save of a host frame [H,W,3] FP64 -> ([H,W,3] uint8 in B,G,R order, (exposure, gain))."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float64)
        assert rgb.shape == (image.height, image.width, 3)
        bgr = np.empty((image.height, image.width, 3), dtype=np.uint8)
        factors = (C.c_double * 2)()
        self._check(self._lib.mcrt_tonemap(self._h, rgb.ctypes.data_as(_dp), C.byref(image), bgr.ctypes.data, factors), "mcrt_tonemap")
        return bgr, (factors[0], factors[1])

    def tonemap_device(self, rgb_ptr, image, bgr_ptr, stream=None):
        """mcrt_tonemap_device on device pointers (e.g. torch tensors' data_ptr()); returns (exposure, gain)."""
        factors = (C.c_double * 2)()
        self._check(self._lib.mcrt_tonemap_device(self._h, C.c_void_p(int(rgb_ptr)), C.byref(image), C.c_void_p(int(bgr_ptr)), factors,
                                                  C.c_void_p(int(stream)) if stream else None), "mcrt_tonemap_device")
        return factors[0], factors[1]

    def emit_photons(self, emissions, caustic_factor, global_seed, shard_index=0, shard_count=1):
        """mcrt_emit_photons[_shard] -> dict(global_=(photons[n,8] f32, keys[n] u64), caustic=(...), paths, rays, kernel_ms)."""
        self._sync_env()
        pe = PhotonEmission()
        self._check(self._lib.mcrt_emit_photons_shard(self._h, float(emissions), float(caustic_factor), int(global_seed),
                                                      int(shard_index), int(shard_count), C.byref(pe)), "mcrt_emit_photons")

        def grab(ptr, kptr, n):
            if n == 0:
                return np.zeros((0, 8), dtype=np.float32), np.zeros(0, dtype=np.uint64)
            return (np.ctypeslib.as_array(ptr, shape=(n * 8,)).reshape(n, 8).copy(),
                    np.ctypeslib.as_array(kptr, shape=(n,)).copy())

        return dict(global_=grab(pe.global_photons, pe.global_keys, pe.global_count),
                    caustic=grab(pe.caustic_photons, pe.caustic_keys, pe.caustic_count),
                    paths=int(pe.emission_paths), rays=int(pe.rays), kernel_ms=pe.kernel_ms)

    def photon_pass_device(self, emissions, caustic_factor, global_seed, bb_min, bb_max, max_photons_per_leaf=200, k_nearest=50,
                           direct_visualization=False):
        """mcrt_photon_pass_device: emission + both maps on the device, installed for the eye pass. Returns the stats dict."""
        self._sync_env()
        st = PhotonPassStats()
        lo, hi = (C.c_double * 3)(*bb_min), (C.c_double * 3)(*bb_max)
        self._check(self._lib.mcrt_photon_pass_device(self._h, float(emissions), float(caustic_factor), int(global_seed), lo, hi,
                                                      int(max_photons_per_leaf), int(k_nearest), 1 if direct_visualization else 0, C.byref(st)),
                    "mcrt_photon_pass_device")
        return st.as_dict()

    def emit_photons_device(self, emissions, caustic_factor, global_seed, shard_index=0, shard_count=1):
        """mcrt_emit_photons_device -> dict(global_=(device pointer, count), caustic=(...), paths, rays, kernel_ms); the lists
        stay in device memory owned by the context."""
        self._sync_env()
        pe = PhotonEmissionDevice()
        self._check(self._lib.mcrt_emit_photons_device(self._h, float(emissions), float(caustic_factor), int(global_seed), int(shard_index),
                                                       int(shard_count), C.byref(pe)), "mcrt_emit_photons_device")
        return dict(global_=(pe.d_global_photons or 0, int(pe.global_count)), caustic=(pe.d_caustic_photons or 0, int(pe.caustic_count)),
                    paths=int(pe.emission_paths), rays=int(pe.rays), kernel_ms=pe.kernel_ms)

    def upload_photons_device(self, d_global, global_count, d_caustic, caustic_count, bb_min, bb_max, max_photons_per_leaf=200, k_nearest=50,
                              direct_visualization=False):
        """mcrt_upload_photons_device: both maps from photon lists in device memory (raw pointers, e.g. tensor.data_ptr())."""
        self._sync_env()
        st = PhotonPassStats()
        lo, hi = (C.c_double * 3)(*bb_min), (C.c_double * 3)(*bb_max)
        self._check(self._lib.mcrt_upload_photons_device(self._h, C.c_void_p(int(d_global)), int(global_count), C.c_void_p(int(d_caustic)),
                                                         int(caustic_count), lo, hi, int(max_photons_per_leaf), int(k_nearest),
                                                         1 if direct_visualization else 0, C.byref(st)), "mcrt_upload_photons_device")
        return st.as_dict()

    def download_map(self, which):
        """mcrt_photon_map_download: host copy (PhotonMap) of installed map 0 (global) / 1 (caustic)."""
        h = C.c_void_p()
        self._check(self._lib.mcrt_photon_map_download(self._h, int(which), C.byref(h)), "mcrt_photon_map_download")
        return PhotonMap._from_handle(h)

    def intersect(self, start, direction):
        self._sync_env()
        start = np.ascontiguousarray(start, dtype=np.float64)
        direction = np.ascontiguousarray(direction, dtype=np.float64)
        n = start.shape[0]
        t = np.empty(n, dtype=np.float64)
        surf = np.empty(n, dtype=np.uint32)
        uv = np.empty((n, 2), dtype=np.float64)
        self._check(self._lib.mcrt_intersect(self._h, n, _ptr(start, C.c_double),
                                             _ptr(direction, C.c_double), _ptr(t, C.c_double),
                                             _ptr(surf, C.c_uint32), _ptr(uv, C.c_double)),
                    "mcrt_intersect")
        return t, surf, uv

    def sampler(self, pixel, index, shuffles, global_seed):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        index = np.ascontiguousarray(index, dtype=np.uint32)
        out = np.empty((pixel.shape[0], 7), dtype=np.float64)
        self._check(self._lib.mcrt_sampler(self._h, pixel.shape[0], _ptr(pixel, C.c_uint32),
                                           _ptr(index, C.c_uint32), int(shuffles), int(global_seed),
                                           _ptr(out, C.c_double)), "mcrt_sampler")
        return out

    def bsdf(self, inputs, consts):
        """mcrt_bsdf: inputs [n][11], consts [10] -> [n][18] (Fresnel / GGX / Oren-Nayar lobe values, include/mcrt.h)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        consts = np.ascontiguousarray(consts, dtype=np.float64)
        out = np.zeros((inputs.shape[0], 18))
        self._check(self._lib.mcrt_bsdf(self._h, inputs.shape[0], _ptr(inputs, C.c_double), _ptr(consts, C.c_double), _ptr(out, C.c_double)), "mcrt_bsdf")
        return out

    def libm(self, fn, a, b=None):
        """mcrt_libm: the device's sincos (fn 0 -> (sin, cos)), sin (1), cos (2), asin (3), atan2 (4: a = y, b = x), sincosf (5: float
        values in, (sin, cos) widened out), pow (6: a ** b) on arrays."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        out0, out1 = np.zeros_like(a), np.zeros_like(a)
        bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
        self._check(self._lib.mcrt_libm(self._h, int(fn), a.size, _ptr(a, C.c_double), _ptr(bb, C.c_double) if bb is not None else None,
                                        _ptr(out0, C.c_double), _ptr(out1, C.c_double)), "mcrt_libm")
        return (out0, out1) if fn in (LIBM_SINCOS, LIBM_SINCOSF) else out0

    def knn(self, which, points, k):
        self._sync_env()
        points = np.ascontiguousarray(points, dtype=np.float64)
        n = points.shape[0]
        cnt = np.empty(n, dtype=np.uint32)
        idx = np.empty((n, k), dtype=np.uint32)
        d2 = np.empty((n, k), dtype=np.float64)
        self._check(self._lib.mcrt_knn(self._h, int(which), n, _ptr(points, C.c_double), int(k),
                                       _ptr(cnt, C.c_uint32), _ptr(idx, C.c_uint32),
                                       _ptr(d2, C.c_double)), "mcrt_knn")
        return cnt, idx, d2

    def close(self):
        if self._h:
            self._lib.mcrt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_rows(cam):
    """Row indices owned by cam's (shard_index, shard_count, shard_rows)."""
    L = lib()
    n = L.mcrt_shard_rows(C.byref(cam), None)
    rows = np.empty(n, dtype=np.uint32)
    if n:
        L.mcrt_shard_rows(C.byref(cam), _ptr(rows, C.c_uint32))
    return rows
