#!/usr/bin/env python3
"""Which device code is in the built library? Extracts the gfx950 code objects of libmcrt_hip.so (llvm-objdump --offloading), disassembles
them and hashes every function's instruction ENCODINGS (position-independent: the words after `// address:` in llvm-objdump's
listing, addresses dropped). Two uses:

  python tools/device_code_hashes.py --check            the library against tests/golden/device_code_hashes.json (what
                                                        tests/test_device_code.py does): which kernels changed since the list was made
  python tools/device_code_hashes.py --write "<note>"   rewrite the list - ONLY after `pytest -m gpu` was green on a GPU box with exactly
                                                        this library; the note names that run (log under profiles/)

The list is the link between "the GPU tests passed" and "this is the code that is in the tree": a kernel edit that was not followed by
a GPU run shows up in the CPU tier as a changed hash. It replaces round 4's by-hand `tools/compare_device_code.py old.s new.s`."""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "libmcrt_hip.so")
LIST = os.path.join(ROOT, "tests", "golden", "device_code_hashes.json")
LLVM = "/opt/rocm/lib/llvm/bin"


def toolchain():
    """What produced the code: hipcc's version line and the compile flags of the device code (monte-carlo-ray-tracer_amd/build.py)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    lines = [l.strip() for l in out.splitlines() if "clang version" in l or l.startswith("HIP version")]
    return " | ".join(lines) or None


def hashes_of(lib=LIB):
    """{mangled function name: sha1 of its instruction encodings} over every gfx950 code object of the library."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    tmp = tempfile.mkdtemp(prefix="mcrt_devcode_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            text = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            name, h = None, None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if name:
                        out[name] = h.hexdigest()
                    name, h = m.group(1), hashlib.sha1()
                    continue
                if name and "//" in line:
                    enc = line.split("//", 1)[1].split(":", 1)
                    if len(enc) == 2:
                        h.update(enc[1].strip().encode())
            if name:
                out[name] = h.hexdigest()
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        return list(names)


def compare(have, want):
    changed = sorted(k for k in have if k in want and have[k] != want[k])
    new = sorted(k for k in have if k not in want)
    gone = sorted(k for k in want if k not in have)
    return changed, new, gone


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--write":
        note = sys.argv[2] if len(sys.argv) > 2 else ""
        if not note:
            raise SystemExit("--write needs a note naming the green GPU run this library was tested by")
        rec = {"validated_by": note, "toolchain": toolchain(), "functions": hashes_of()}
        with open(LIST, "w") as f:
            json.dump(rec, f, indent=0, sort_keys=True)
        print("wrote %s: %d functions" % (os.path.relpath(LIST, ROOT), len(rec["functions"])))
        return 0
    rec = json.load(open(LIST))
    changed, new, gone = compare(hashes_of(), rec["functions"])
    print("list: %s (%d functions, validated by: %s)" % (os.path.relpath(LIST, ROOT), len(rec["functions"]), rec["validated_by"]))
    for title, names in (("changed", changed), ("new", new), ("gone", gone)):
        for n in demangle(names):
            print("  %s: %s" % (title, n[:170]))
    print("identical: %d, changed: %d, new: %d, gone: %d" % (len(rec["functions"]) - len(changed) - len(gone), len(changed), len(new), len(gone)))
    return 1 if (changed or new or gone) else 0


if __name__ == "__main__":
    sys.exit(main())
