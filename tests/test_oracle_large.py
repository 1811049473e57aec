"""The oracle against the reference on BASELINE configs[2] at full size: two full-width rows of
metal_bunnies (synthetic stand-in bunny, tests/large/make_synthetic.py) at 1920x1080 @ 1024 spp over the
reference's own quaternary-SAH BVH (169 162 nodes, 491 592 triangles + 1 sphere). Runs wherever
oracle/_ref holds the reference binary and the scene copy (build container, GPU box); about 15 s on 8 cores."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "large"))


def test_synthetic_bunny_is_reproducible(tmp_path):
    import hashlib
    import make_large
    import make_synthetic
    p = str(tmp_path / "bunny.obj")
    assert make_synthetic.write_bunny(p) == (40962, 81920)
    assert hashlib.md5(open(p, "rb").read()).hexdigest() == make_large.BUNNY_MD5


def test_oracle_c3_rows_equal_reference(pkg, oracle):
    import make_large
    p = make_large.ensure_c3_image()
    if p is None:
        pytest.skip("oracle/_ref (reference binary + metal_bunnies scene copy) not on this machine")
    img = pkg.SceneImage(p)
    c3 = make_large.C3
    out, info = oracle.render(img, img.camera, make_large.SEED, pkg.INTEGRATOR_PATH_TRACER, rows=c3["rows"])
    ref = np.fromfile(c3["golden"]).reshape(out.shape)
    np.testing.assert_array_equal(out, ref)
    assert info["node_tests"] / info["rays"] > 30  # a deep tree: ~46 box tests and ~7 primitive tests per ray
