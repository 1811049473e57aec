"""The device code in the tree is the device code the GPU tests last ran on.

tests/golden/device_code_hashes.json lists, per gfx950 function of libmcrt_hip.so, a hash of its instruction encodings; it is written
(tools/device_code_hashes.py --write) only after a green `pytest -m gpu` on a GPU box with exactly that library, and names that run.
This CPU-tier test disassembles the library the build just made and compares: a kernel that was edited - or whose code changed because
a header it includes did - without a GPU run since fails here by name. (Round 4 showed this by hand with tools/compare_device_code.py
after refactoring product headers for the host emulation.)"""
import json
import os

import pytest

from conftest import ROOT


def test_built_kernels_are_the_ones_the_gpu_tests_ran(pkg):
    import importlib.util
    spec = importlib.util.spec_from_file_location("device_code_hashes", os.path.join(ROOT, "tools", "device_code_hashes.py"))
    dch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dch)
    if not os.path.exists(os.path.join(dch.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump on this machine")
    rec = json.load(open(dch.LIST))
    tc = dch.toolchain()
    if tc != rec.get("toolchain"):
        pytest.skip("another compiler than the list was made with (%r vs %r): hashes are not comparable" % (tc, rec.get("toolchain")))
    pkg.lib()  # (built in tree by the fixture)
    have = dch.hashes_of()
    changed, new, gone = dch.compare(have, rec["functions"])
    assert len(have) > 300
    names = dch.demangle(changed + new + gone)
    assert not (changed or new or gone), (
        "device code differs from the list validated by %r - %d changed, %d new, %d gone, e.g. %s. Run `pytest -m gpu` on a GPU box, then "
        "`python tools/device_code_hashes.py --write \"<that run>\"`." % (rec["validated_by"], len(changed), len(new), len(gone), names[:4]))
