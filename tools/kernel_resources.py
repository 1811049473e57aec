#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS table of a gfx950 assembly file produced by
`hipcc --offload-arch=gfx950 ... -save-temps=obj -c x.hip -o /tmp/x.o` (the .s next to the object).

  python tools/kernel_resources.py /tmp/mcrt_hip-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]

Also counts the instructions of each kernel body by class (VALU f64, other VALU, SALU, LDS, VMEM, scratch),
which is how instruction-stream changes are judged in the GPU-less build container."""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    s = open(path).read()
    meta = s[s.index("amdhsa.kernels"):]
    ents = meta.split("  - .agpr_count")[1:]
    names = [re.search(r"\.name:\s+(\S+)", e).group(1) for e in ents]
    dm = demangle(names)
    # instruction histogram per kernel: text between "<name>:" and ".Lfunc_end"
    bodies = {}
    for n in names:
        m = re.search(r"^" + re.escape(n) + r":[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M)
        bodies[n] = m.group(1) if m else ""
    for e, n in zip(ents, names):
        d = dm[n]
        if flt and flt not in d:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", e).group(1))
        cls = dict(f64=0, valu=0, salu=0, lds=0, vmem=0, scratch=0, total=0)
        for line in bodies[n].splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":") or t[0] == "s_nop" or t[0] == "s_endpgm":
                continue
            op = t[0]
            cls["total"] += 1
            if op.startswith("scratch_"):
                cls["scratch"] += 1
            elif op.startswith("v_") and "f64" in op:
                cls["f64"] += 1
            elif op.startswith("v_"):
                cls["valu"] += 1
            elif op.startswith("s_"):
                cls["salu"] += 1
            elif op.startswith("ds_"):
                cls["lds"] += 1
            elif op.startswith(("global_", "flat_", "buffer_")):
                cls["vmem"] += 1
        print("%-100s vgpr=%3d spill=%3d sgpr=%3d scratch=%4d B | static instr %6d: f64 %5d valu %5d salu %5d lds %4d vmem %4d scratch %4d"
              % (d[:100], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("private_segment_fixed_size"),
                 cls["total"], cls["f64"], cls["valu"], cls["salu"], cls["lds"], cls["vmem"], cls["scratch"]))


if __name__ == "__main__":
    main()
