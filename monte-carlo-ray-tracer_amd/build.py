"""In-tree build of libmcrt_hip.so (gfx950 only) with hipcc. No JIT cache, no pip install: the built
library sits next to its sources so that it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(CSRC, "libmcrt_hip.so")
# The opt-in TOLERANCE build (MCRT_TOLERANCE_BUILD=1 selects it at import, monte-carlo-ray-tracer_amd/__init__.py): the kernels' translation unit
# compiled with -ffp-contract=fast (mul + add pairs fuse into FP64 fma: half the instructions of a dot product) and -DMCRT_PLATFORM_LIBM
# (the platform's sin / cos / asin / atan2 / pow instead of the restated glibc routines). Frames then agree with the reference within
# BASELINE.json's 1e-4 relative bar (measured: tests/test_gpu_tolerance_build.py, bench.py's *_tol legs) instead of bit for bit. The exact
# build stays the default, the headline and what every parity test runs.
LIB_TOL = os.path.join(CSRC, "libmcrt_hip_tol.so")
TOL_FLAGS = ["-ffp-contract=fast", "-DMCRT_PLATFORM_LIBM", "-DMCRT_TOLERANCE_BUILD"]
RENDER_BIN = os.path.join(HOST, "mcrt_render")

# -ffp-contract=off: the CPU reference is compiled by g++ for baseline x86-64 (no FMA contraction);
# per-pixel FP64 parity needs the same rounding sequence on the GPU.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
# MCRT_PLATFORM_LIBM=1: a build without the restated glibc routines and their tables (csrc/mcrt_libm.hpp "BUILD SWITCH", NOTICE): the
# platform's libm instead - frames within 1e-12 of the reference's rather than its bits
if os.environ.get("MCRT_PLATFORM_LIBM") == "1":
    HIPCC_FLAGS.append("-DMCRT_PLATFORM_LIBM")
# MCRT_EXACT_PHOTON_DIR=1: Photon::dir's sine / cosine pairs by the restated sincosf (glibc's bits) instead of the platform's sinf / cosf
# (csrc/mcrt_integrator.hpp photonDirection: +8.8 % on a C5 frame, nothing a test can see)
if os.environ.get("MCRT_EXACT_PHOTON_DIR") == "1":
    HIPCC_FLAGS.append("-DMCRT_EXACT_PHOTON_DIR")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def sources():
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".hpp"))]
    src.append(os.path.join(HERE, "..", "include", "mcrt.h"))
    return src


# the translation units that hold kernels of the render path: the tolerance library has its own objects of these
KERNEL_TUS = ("mcrt_hip.hip", "mcrt_hip_lean.hip")
TUS = ["mcrt_hip.hip", "mcrt_hip_lean.hip", "mcrt_octree_gpu.hip", "mcrt_sah_gpu.hip", "mcrt_output.hip", "mcrt_multi.hip", "mcrt_image.cpp", "mcrt_octree.cpp", "mcrt_bvh.cpp"]
OBJ = os.path.join(CSRC, "_obj")


def _deps_of(depfile):
    """Prerequisites listed in a make-style dependency file written by `hipcc -MD -MF`."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    return [t for t in text.split(":", 1)[1].split() if t] if ":" in text else None


def _stale(obj, depfile):
    deps = _deps_of(depfile)
    if deps is None or not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _compile_jobs(force, variants):
    """hipcc command lines for the stale objects of the given variants (False = exact, True = tolerance). The tolerance library only
    has its own object for the kernels' translation unit; every other object is the exact build's."""
    hipcc = _hipcc()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    for tu in TUS:
        for tol in sorted(set(bool(v) and tu in KERNEL_TUS for v in variants)):
            stem = os.path.join(OBJ, tu + (".tol" if tol else ""))
            obj, dep = stem + ".o", stem + ".d"
            if force or _stale(obj, dep):
                f = [x for x in flags if x != "-ffp-contract=off"] + TOL_FLAGS if tol else flags
                jobs.append([hipcc] + f + ["-MD", "-MF", dep, "-c", os.path.join(CSRC, tu), "-o", obj])
    return jobs


def _objects(tolerance):
    return [os.path.join(OBJ, tu + (".tol" if tolerance and tu in KERNEL_TUS else "") + ".o") for tu in TUS]


def build_lib(force=False, verbose=True, tolerance=False, both=False):
    """One object per translation unit (compiled side by side, only the stale ones — hipcc's own -MD dependency files
    decide), then one link. The objects stay under csrc/_obj/ (git-ignored). tolerance=True: libmcrt_hip_tol.so - the kernels'
    translation unit (mcrt_hip.hip) recompiled with TOL_FLAGS, linked with the exact build's other objects (builders, output stage).
    both=True: the two libraries, their two compiles of mcrt_hip.hip (minutes each) side by side."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ, exist_ok=True)
    variants = (False, True) if both else (bool(tolerance),)
    jobs = _compile_jobs(force, variants)

    def run(cmd):
        # hipcc reads a translation unit's headers twice (device pass, then host pass), minutes apart for mcrt_hip.hip: a header saved in
        # between gives ONE object whose kernels and launch code disagree about a struct's layout (round 6: a memory fault on the GPU that
        # no CPU test could see). A compile whose dependencies changed while it ran is therefore done again.
        for attempt in range(3):
            t_start = time.time()
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            dep = cmd[cmd.index("-MF") + 1] if "-MF" in cmd else None
            deps = _deps_of(dep) if dep else None
            if not deps or all(os.path.exists(d) and os.path.getmtime(d) < t_start for d in deps):
                return
            if verbose:
                print("[build] a dependency changed during the compile: again", flush=True)
        raise RuntimeError("sources kept changing while %s was being compiled" % cmd[-3])

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    for tol in variants:
        lib, objs = (LIB_TOL if tol else LIB), _objects(tol)
        if jobs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
            run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB_TOL if tolerance and not both else LIB


def build_host(force=False, verbose=True):
    """C++ host driver (stand-alone renderer on top of the C ABI): host/mcrt_render."""
    src = os.path.join(HOST, "mcrt_render.cpp")
    if not os.path.exists(src):
        return None
    if not force and _newer(RENDER_BIN, [src, LIB]):
        return RENDER_BIN
    cmd = ["g++", "-std=c++17", "-O2", "-o", RENDER_BIN, src, "-L" + CSRC, "-lmcrt_hip", "-Wl,-rpath," + CSRC,
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return RENDER_BIN


def build_all(force=False):
    build_lib(force, both=os.environ.get("MCRT_SKIP_TOLERANCE_BUILD") != "1")
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
