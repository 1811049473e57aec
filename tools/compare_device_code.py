#!/usr/bin/env python3
"""Are the kernels of two builds the same instructions? Compares, function by function, the gfx950 assembly of two
`hipcc --offload-arch=gfx950 ... -save-temps=obj -c mcrt_hip.hip` runs (the *-hip-amdgcn-amd-amdhsa-gfx950.s files), with labels and
comments stripped. Used when the GPU is not at hand: a change that is meant to leave the device code alone (a refactor of host code, a
macro, code moved between headers, a new OPTIONAL kernel instance) is shown to leave every existing kernel instruction-identical to
the tree the GPU tests last ran on.

    python tools/compare_device_code.py old.s new.s"""
import hashlib
import re
import subprocess
import sys


def bodies(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'^(_ZN[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        body = re.sub(r'\.L[A-Za-z0-9_]+', 'L', m.group(2))
        body = '\n'.join(re.sub(r';.*$', '', l).rstrip() for l in body.splitlines() if not l.strip().startswith((';', '.')))
        out[m.group(1)] = hashlib.md5(body.encode()).hexdigest()
    return out


def main():
    a, b = bodies(sys.argv[1]), bodies(sys.argv[2])
    demangle = lambda names: subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    same = [k for k in a if k in b and a[k] == b[k]]
    diff = [k for k in a if k in b and a[k] != b[k]]
    print("functions in %s: %d; identical in %s: %d; different: %d; gone: %d; new: %d" %
          (sys.argv[1], len(a), sys.argv[2], len(same), len(diff), sum(1 for k in a if k not in b), sum(1 for k in b if k not in a)))
    for title, names in (("different", diff), ("gone", [k for k in a if k not in b]), ("new", [k for k in b if k not in a])):
        for n in demangle(names):
            print("  %s: %s" % (title, n[:160]))
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
