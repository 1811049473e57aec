"""In-tree build of libmcrt_hip.so (gfx950 only) with hipcc. No JIT cache, no pip install: the built
library sits next to its sources so that it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(CSRC, "libmcrt_hip.so")
RENDER_BIN = os.path.join(HOST, "mcrt_render")

# -ffp-contract=off: the CPU reference is compiled by g++ for baseline x86-64 (no FMA contraction);
# per-pixel FP64 parity needs the same rounding sequence on the GPU.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


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


def build_lib(force=False, verbose=True):
    deps = sources()
    if not force and _newer(LIB, deps):
        return LIB
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB, os.path.join(CSRC, "mcrt_hip.hip"), os.path.join(CSRC, "mcrt_octree_gpu.hip"), os.path.join(CSRC, "mcrt_output.hip"), os.path.join(CSRC, "mcrt_multi.hip"),
                                         os.path.join(CSRC, "mcrt_image.cpp"), os.path.join(CSRC, "mcrt_octree.cpp"), os.path.join(CSRC, "mcrt_bvh.cpp")]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


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
    build_lib(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
