#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel in the BUILT library, read from the code objects' own metadata
(`llvm-readelf --notes` on the gfx950 objects that `llvm-objdump --offloading` extracts from libmcrt_hip.so).

  python tools/kernel_spill_table.py                       markdown table on stdout (committed as profiles/rNN_kernel_resources.md)
  python tools/kernel_spill_table.py --check               default-path kernels against tests/golden/kernel_spill_budget.json
  python tools/kernel_spill_table.py --write-budget        rewrite that budget from the built library

The budget is a ceiling per default-path kernel family: tests/test_kernel_resources.py (CPU tier) fails when an edit makes one of them
spill more than the list allows - spills are how a kernel of this library gets slower without any test noticing."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "libmcrt_hip.so")
BUDGET = os.path.join(ROOT, "tests", "golden", "kernel_spill_budget.json")
LLVM = "/opt/rocm/lib/llvm/bin"
# the kernel instances the default options launch (launchRender / launchWavefront / the photon pass, csrc/mcrt_hip.hip), by demangled-name prefix
DEFAULT_PATH = ("renderKernelFlatK<768>", "renderKernelSM<false, false, false, 512>", "wfTraceKernel<(anonymous namespace)::PoolRays, false, 3>",
                "wfTraceKernel<(anonymous namespace)::PoolRays, false, 1>", "wfShadeKernel<false>", "wfShadeKernel<true>", "wfKnnKernel<true, 4>",
                "renderKernelPM<false, false, 1024, 4>", "renderKernelPM<false, true, 1024, 4>", "emitKernel<false>", "emitKernel<true>", "sampleResolveKernel",
                # the lean instances (csrc/mcrt_hip_lean.hip: scenes without rough / conductor materials - hexagon_room, water_caustics - run these)
                "lean::renderKernelFlatK<512>", "lean::renderKernelPM<false, true, 1024, 4>", "lean::renderKernelSM<false, false, false, 512>",
                "lean::wfShadeKernel<false>", "lean::wfShadeKernel<true>", "lean::emitKernel<false>", "lean::emitKernel<true>", "lean::wfKnnKernel<true, 4>")


def short(name):
    k = name.replace("(anonymous namespace)::", "").replace("mcrt::", "")
    if k.startswith("void "):
        k = k[5:]
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return k[:i].strip()
    return k.strip()


def kernels_of(lib=LIB):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_wg}] over every gfx950 code object of the library."""
    tmp = tempfile.mkdtemp(prefix="mcrt_res_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for ent in notes.split("  - .agpr_count:")[1:]:
                ent = ".agpr_count:" + ent
                g = lambda k, d=0: int(m.group(1)) if (m := re.search(r"\." + k + r":\s+(\d+)", ent)) else d
                name = re.search(r"\.name:\s+(\S+)", ent).group(1)
                out.append(dict(mangled=name, vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                                sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                                max_wg=g("max_flat_workgroup_size")))
        names = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in out), capture_output=True, text=True).stdout.splitlines()
        for k, n in zip(out, names):
            k["name"] = short(n)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def is_default(name):
    n = name.replace("(anonymous namespace)::", "")
    return any(n.startswith(d.replace("(anonymous namespace)::", "")) for d in DEFAULT_PATH)


def main():
    ks = kernels_of()
    if "--write-budget" in sys.argv:
        b = {k["name"]: {"vgpr_spill": k["vgpr_spill"], "sgpr_spill": k["sgpr_spill"], "scratch": k["scratch"]} for k in ks if is_default(k["name"])}
        json.dump({"note": "ceilings per default-path kernel (tools/kernel_spill_table.py --write-budget); a kernel may spill less, never more", "kernels": b},
                  open(BUDGET, "w"), indent=1, sort_keys=True)
        print("wrote %s (%d kernels)" % (BUDGET, len(b)))
        return 0
    if "--check" in sys.argv:
        want = json.load(open(BUDGET))["kernels"]
        have = {k["name"]: k for k in ks}
        bad = ["%s: missing from the library" % n for n in want if n not in have]
        for n, w in want.items():
            k = have.get(n)
            if k and (k["vgpr_spill"] > w["vgpr_spill"] or k["scratch"] > w["scratch"]):
                bad.append("%s: %d spilled VGPRs / %d B scratch, budget %d / %d" % (n, k["vgpr_spill"], k["scratch"], w["vgpr_spill"], w["scratch"]))
        print("\n".join(bad) if bad else "ok: %d default-path kernels within their spill budgets" % len(want))
        return 1 if bad else 0
    print("# Kernel resources of the built `libmcrt_hip.so` (code-object metadata, `llvm-readelf --notes`)\n")
    print("`tools/kernel_spill_table.py`; **bold** = the instances the default options launch. waves/SIMD = min(8, 512 / allocated VGPRs) capped by the launch bound.\n")
    print("| kernel | max WG | VGPR | AGPR | SGPR | spilled VGPR | spilled SGPR | scratch B/lane | static LDS B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k in sorted(ks, key=lambda k: (not is_default(k["name"]), k["name"])):
        n = ("**`%s`**" if is_default(k["name"]) else "`%s`") % k["name"][:110]
        print("| %s | %d | %d | %d | %d | %d | %d | %d | %d |" % (n, k["max_wg"], k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"], k["lds"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
