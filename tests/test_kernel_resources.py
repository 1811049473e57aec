"""CPU tier: the default-path kernels of the BUILT library spill no more than tests/golden/kernel_spill_budget.json allows
(tools/kernel_spill_table.py reads the code objects' metadata: vgpr_spill_count, private_segment_fixed_size). A kernel that starts to
spill more gets slower without any parity test noticing - round 5's renderKernelPM moved +-10 % on unrelated one-line edits."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_spill_table", os.path.join(ROOT, "tools", "kernel_spill_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="llvm-readelf not in this image")
def test_default_path_kernels_stay_within_their_spill_budgets():
    t = _tool()
    if not os.path.exists(t.LIB):
        pytest.skip("libmcrt_hip.so not built")
    want = json.load(open(t.BUDGET))["kernels"]
    have = {k["name"]: k for k in t.kernels_of()}
    assert want, "empty budget"
    for name, w in want.items():
        assert name in have, "default-path kernel %s is not in the library" % name
        k = have[name]
        assert k["vgpr_spill"] <= w["vgpr_spill"], "%s spills %d VGPRs (budget %d)" % (name, k["vgpr_spill"], w["vgpr_spill"])
        assert k["scratch"] <= w["scratch"], "%s uses %d B of scratch per lane (budget %d)" % (name, k["scratch"], w["scratch"])
    # the trace kernel of the pipeline must not spill at all: it runs at 4 waves per SIMD on 128 VGPRs
    for name, k in have.items():
        if name.startswith("wfTraceKernel<PoolRays, false"):
            assert k["vgpr_spill"] == 0 and k["scratch"] == 0, name


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not in this image")
@pytest.mark.parametrize("switch,kernel", [("-DMCRT_EXACT_PHOTON_DIR", "renderKernelPM<false, false, 1024, 4>"),
                                           ("-DMCRT_PLATFORM_LIBM", "wfShadeKernel<false>")])
def test_build_switches_compile_for_gfx950(switch, kernel):
    """The two build-time switches of csrc (monte-carlo-ray-tracer_amd/build.py reads them from the environment) produce device code:
    one kernel that contains the switched code is compiled for gfx950 with each (tools/one_kernel.sh; the whole tolerance library, which
    is built with -DMCRT_PLATFORM_LIBM, has its own GPU test: tests/test_gpu_tolerance_build.py)."""
    import subprocess
    p = subprocess.run([os.path.join(ROOT, "tools", "one_kernel.sh"), kernel, switch], capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert "error" not in out.lower(), out[-2000:]
    assert kernel.split("<")[0] in out and "vgpr=" in out, out[-2000:]


def test_every_lean_kernel_id_has_an_instance(pkg):
    """csrc/mcrt_lean.hpp lists the lean instances by id; csrc/mcrt_hip_lean.hip must hold one for each (mcrt_lean_kernel returns the
    address hipLaunchKernel takes - a host stub, valid without a GPU), and none beyond the list."""
    import ctypes as C
    import re
    hdr = open(os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_lean.hpp")).read()
    body = hdr[hdr.index("enum McrtLeanKernelId {"):hdr.index("MCRT_LEAN_COUNT")]
    ids = re.findall(r"^\s*(MCRT_LEAN_[A-Z0-9_]+)", body, re.M)
    assert len(ids) >= 10
    L = pkg.lib()
    L.mcrt_lean_kernel.argtypes = [C.c_int]
    L.mcrt_lean_kernel.restype = C.c_void_p
    got = [L.mcrt_lean_kernel(i) for i in range(len(ids))]
    assert all(got), "no instance for %s" % [n for n, g in zip(ids, got) if not g]
    assert len(set(got)) == len(got)
    assert L.mcrt_lean_kernel(len(ids)) is None and L.mcrt_lean_kernel(-1) is None
