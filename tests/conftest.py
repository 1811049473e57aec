import ctypes as C
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(TESTS, ".."))
GOLDEN = os.path.join(TESTS, "golden")
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (the HIP path has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _dirty_device_memory(request):
    """GPU tests run on memory that is NOT zero: before each of them a few GB are filled with 0xFF bytes (NaN as doubles, ~4e9 as
    indices) and handed back, so that whatever the library allocates next is dirty — as it is on a GPU that other processes or
    contexts use too. A result that depends on fresh memory being zero (an unwritten word read as 0) then fails here, on the
    single-process box, instead of only when the device is shared."""
    if "gpu" in request.keywords and _has_gpu():
        import torch
        try:
            junk = [torch.full((1 << 30,), 0xFF, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        except Exception:
            pass
    yield


@pytest.fixture(scope="session")
def pkg():
    """The product binding. The shared library must already be built (python __graft_entry__.py build)."""
    m = importlib.import_module("monte-carlo-ray-tracer_amd")
    if not os.path.exists(m.LIB_PATH):
        importlib.import_module("monte-carlo-ray-tracer_amd.build").build_lib()
    m.lib()
    return m


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="session")
def emu():
    return load_emu()


@pytest.fixture(scope="session")
def wave_emu():
    """Host build of the product's WAVE-cooperative photon search (tests/emu/wave_knn_emu.cpp: csrc/mcrt_waveknn.hpp unchanged, one
    wavefront = 64 fibers, tests/emu/wave_emu.hpp) — test harness only."""
    src = os.path.join(TESTS, "emu", "wave_knn_emu.cpp")
    out = os.path.join(TESTS, "emu", "_build", "libwave_knn_emu.so")
    csrc = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc")
    deps = [src, os.path.join(TESTS, "emu", "wave_emu.hpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fno-inline", "-ffp-contract=off", "-fPIC", "-shared", "-o", tmp, src])
        os.replace(tmp, out)
    L = C.CDLL(out)
    L.wemu_knn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wemu_knn.restype = C.c_int
    return L


@pytest.fixture(scope="session")
def wave_walk_emu():
    """Host build of the trace kernels' wave-synchronous walk with the shared leaf step (tests/emu/wave_walk_emu.cpp:
    csrc/mcrt_sharedleaf.hpp unchanged, 64 rays per emulated wavefront) — test harness only."""
    src = os.path.join(TESTS, "emu", "wave_walk_emu.cpp")
    out = os.path.join(TESTS, "emu", "_build", "libwave_walk_emu.so")
    csrc = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc")
    deps = [src, os.path.join(TESTS, "emu", "wave_emu.hpp"), os.path.join(TESTS, "emu", "mcrt_emu.cpp")] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fno-inline", "-ffp-contract=off", "-fPIC", "-shared", "-o", tmp, src])
        os.replace(tmp, out)
    L = C.CDLL(out)
    L.wemu_intersect.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wemu_intersect.restype = C.c_int
    L.wemu_estimate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wemu_estimate.restype = C.c_int
    return L


def _wave_kernel_emu(lib_name, extra_flags=()):
    src = os.path.join(TESTS, "emu", "wave_kernel_emu.cpp")
    out = os.path.join(TESTS, "emu", "_build", lib_name)
    csrc = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc")
    deps = [src, os.path.join(TESTS, "emu", "wave_emu.hpp"), os.path.join(TESTS, "emu", "mcrt_emu.cpp")] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fno-inline", "-ffp-contract=off", "-fPIC", "-shared"] + list(extra_flags) + ["-o", tmp, src])
        os.replace(tmp, out)
    L = C.CDLL(out)
    vp = C.c_void_p
    L.wemu_trace_kernel.argtypes = [vp, C.c_uint64, vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, vp, vp, vp, vp]
    L.wemu_trace_kernel.restype = C.c_int
    L.wemu_render.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_int, C.c_int, C.c_uint32, vp, vp, vp]
    L.wemu_render.restype = C.c_int
    L.wemu_render_pipeline.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]
    L.wemu_render_pipeline.restype = C.c_int
    L.wemu_emit.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp]
    L.wemu_emit.restype = C.c_int
    L.wemu_set_shuffle.argtypes = [C.c_uint64]
    L.wemu_set_shuffle.restype = None
    return L


@pytest.fixture(scope="session")
def wave_kernel_emu():
    """Host build of the product's KERNELS (tests/emu/wave_kernel_emu.cpp: csrc/mcrt_kernels.hpp unchanged, workgroups of emulated
    wavefronts with __syncthreads, LDS and the launch geometry) — test harness only."""
    return _wave_kernel_emu("libwave_kernel_emu.so")


@pytest.fixture(scope="session")
def wave_kernel_emu_lean():
    """The same kernels compiled as csrc/mcrt_hip_lean.hip compiles them: without Oren-Nayar, GGX and conductor Fresnel
    (MCRT_MAT_FEATURES_OFF = MCRT_MAT_ROUGH | MCRT_MAT_ROUGH_SPECULAR | MCRT_MAT_COMPLEX_IOR, csrc/mcrt_shade.hpp) — test harness only."""
    return _wave_kernel_emu("libwave_kernel_emu_lean.so", ["-DMCRT_MAT_FEATURES_OFF=67u"])


def load_emu():
    """Host build of the product's per-lane device code (tests/emu/mcrt_emu.cpp) — test harness only."""
    src = os.path.join(TESTS, "emu", "mcrt_emu.cpp")
    out = os.path.join(TESTS, "emu", "_build", "libmcrt_emu.so")
    csrc = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())  # (several pytest-xdist workers may get here at once: build aside, rename into place)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", tmp, src])
        os.replace(tmp, out)
    L = C.CDLL(out)
    vp = C.c_void_p
    L.emu_render.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
    L.emu_intersect.argtypes = [vp, C.c_uint64, vp, vp, C.c_int, vp, vp, vp]
    L.emu_render_sm.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
    L.emu_render_wf.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.emu_render_wf_pm.argtypes = [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.emu_emit_photons.argtypes = [vp, C.c_double, C.c_double, C.c_uint32, C.c_int, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp]
    L.emu_octree_build.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.emu_octree_desc.argtypes = [vp]
    L.emu_octree_desc.restype = vp
    L.emu_octree_free.argtypes = [vp]
    L.emu_sampler.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.emu_sampler_restore.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.emu_shade_rec.argtypes = [vp, vp]
    L.emu_shade_rec.restype = C.c_int
    L.emu_render_wf_film.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.emu_film_resolve.argtypes = [vp, C.c_uint64, vp]
    L.emu_film_resolve.restype = None
    L.emu_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint64, vp]
    L.emu_plan.restype = None
    L.emu_plan_mega.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, vp]
    L.emu_plan_mega.restype = None
    L.emu_plan_pool_slots.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.emu_plan_pool_slots.restype = C.c_uint64
    L.emu_tonemap.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, vp, vp]
    L.emu_flat_cull.argtypes = [vp, C.c_uint64, vp, vp, vp]
    L.emu_qstep_check.argtypes = [vp, C.c_uint64, vp, vp, vp]
    L.emu_knn.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp]
    L.emu_knn_cap.argtypes = [vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.emu_knn_cap.restype = C.c_int
    L.emu_trace_counts.argtypes = [vp, C.c_uint64, vp, vp, C.c_int, vp]
    L.emu_set_defer.argtypes = [C.c_int]
    return L


def golden_path(name):
    return os.path.join(GOLDEN, name)


def load_radiance(render):
    r0, r1 = render["rows"]
    return np.fromfile(golden_path(render["file"]), dtype=np.float64).reshape(r1 - r0, render["width"], 3)


def camera_for(image, render):
    cam = image.camera
    cam.width, cam.height, cam.sqrtspp = render["width"], render["height"], render["sqrtspp"]
    return cam


def rel_error(a, ref):
    """per-pixel, per-channel |a-ref| / max(|ref|, 1e-3)  (SURVEY.md §8(d) parity metric)."""
    return np.abs(a - ref) / np.maximum(np.abs(ref), 1e-3)


def check_hits_against_reference(oracle, img, kat_dir, t, surf, uv):
    """t must equal the reference's bits for every ray. The surface must be the reference's unless two
    different surfaces are hit at exactly the same t: the reference keeps whichever its best-first heap
    order tested first (bvh.cpp:100), the HIP path keeps the lowest surface index (order independent).
    Returns the number of such ties."""
    rays = np.fromfile(os.path.join(kat_dir, "isect_rays.f64")).reshape(-1, 6)
    t_ref = np.fromfile(os.path.join(kat_dir, "isect_t.f64"))
    s_ref = np.fromfile(os.path.join(kat_dir, "isect_surface.u32"), dtype=np.uint32)
    uv_ref = np.fromfile(os.path.join(kat_dir, "isect_uv.f64")).reshape(-1, 2)
    np.testing.assert_array_equal(t, t_ref)
    diff = np.nonzero(surf != s_ref)[0]
    for i in diff:
        o, d = rays[i:i + 1, :3].copy(), rays[i:i + 1, 3:].copy()
        assert surf[i] < s_ref[i], "ray %d: tie must go to the lowest surface index" % i
        assert oracle.single_surface_t(img, surf[i], o, d)[0] == t_ref[i], "ray %d: surface %d is not hit at the reference t" % (i, surf[i])
        assert oracle.single_surface_t(img, s_ref[i], o, d)[0] == t_ref[i]
    same = surf == s_ref
    np.testing.assert_array_equal(uv[same], uv_ref[same])
    assert len(diff) <= max(3, len(t) // 500)
    return len(diff)


def assert_same_octree(a, b):
    """Two photon maps (PhotonMap.arrays() dicts) are the same tree: identical octant arrays, and in every leaf the same
    photons (compared as sets of 32-byte records: the order inside a leaf is builder specific and no query depends on it)."""
    for k in ("bounds", "start", "contained", "next", "leaf"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["photons"].shape == b["photons"].shape
    pa, pb = a["photons"].view(np.uint32), b["photons"].view(np.uint32)  # bit patterns: NaN-safe, -0.0 != 0.0
    for i in np.nonzero(a["leaf"])[0]:
        s, n = int(a["start"][i]), int(a["contained"][i])
        ra = np.sort(np.ascontiguousarray(pa[s:s + n]).view("V32").ravel())
        rb = np.sort(np.ascontiguousarray(pb[s:s + n]).view("V32").ravel())
        assert ra.tobytes() == rb.tobytes(), "leaf %d holds different photons" % i


def sort_by_key(photons, keys):
    order = np.argsort(keys, kind="stable")
    return photons[order], keys[order]


def photon_set_bytes(photons):
    """The photon list as a sorted array of opaque 32-byte records (order-free exact comparison)."""
    return np.sort(np.ascontiguousarray(photons, dtype=np.float32).view("V32").ravel())


def host_libm_is_the_restated_one():
    """True where the ORACLE's libm calls return what csrc/mcrt_libm.hpp restates: x86-64 glibc 2.35 on a CPU with FMA + AVX2 (the IFUNC
    variants that were transcribed). Anywhere else the oracle itself differs from the reference in last bits, and tests that compare the
    GPU with the ORACLE (not with reference-made fixtures, which do not depend on the host) fall back to a tolerance."""
    import platform
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    libc, ver = platform.libc_ver()
    return platform.machine() == "x86_64" and libc == "glibc" and ver.startswith("2.35") and "fma" in flags and "avx2" in flags


def assert_oracle_bits(got, want, what="", rel=2e-6):
    """Float arrays of the GPU (or of device code run on the host) against the oracle's: the same bits where the host's libm is the
    restated one, within `rel` elsewhere (photon records are FP32: angles and positions rounded from FP64 values that differ by an ulp)."""
    got, want = np.asarray(got), np.asarray(want)
    if host_libm_is_the_restated_one():
        np.testing.assert_array_equal(got.view(np.uint32 if got.dtype == np.float32 else np.uint64), want.view(np.uint32 if want.dtype == np.float32 else np.uint64),
                                      err_msg=what)
    else:
        np.testing.assert_allclose(got, want, rtol=rel, atol=rel, err_msg=what + " (host libm is not x86-64 glibc 2.35 with FMA: tolerance instead of bits)")
