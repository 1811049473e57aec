"""The out-of-LDS traversal path on a real BVH: spaceship.json (quaternary SAH built by the reference,
23 187 nodes / 68 760 triangles / 354 emissive triangles): top 512 nodes staged in LDS, everything else
read from HBM/L2. Fixture made by tests/large/make_large.py (reference output, not committed: 17 MB)."""
import os

import numpy as np
import pytest

from conftest import ROOT, rel_error

pytestmark = pytest.mark.gpu
IMAGES = os.path.join(ROOT, "oracle", "_ref", "images")


@pytest.fixture(scope="module")
def spaceship(pkg):
    p = os.path.join(IMAGES, "spaceship.mcrt")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/images/spaceship.mcrt not built (python __graft_entry__.py build in the build container)")
    return pkg.SceneImage(p)


def test_spaceship_matches_reference(pkg, spaceship):
    ctx = pkg.Context(0)
    ctx.upload_image(spaceship)
    cam = spaceship.camera
    cam.width, cam.height, cam.sqrtspp = 480, 270, 2
    out, st = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PATH_TRACER)
    ref = np.fromfile(os.path.join(IMAGES, "spaceship.480x270_s2.f64")).reshape(270, 480, 3)
    rel = rel_error(out, ref).max(axis=2)
    bad = int((rel > 1e-4).sum())
    print("spaceship: max rel %.3e, outliers %d / %d, %.1f Mray/s, %.2f rays/path" %
          (rel.max(), bad, rel.size, st["rays"] / st["kernel_ms"] / 1e3, st["rays"] / st["paths"]))
    assert bad <= int(0.002 * rel.size)
    ctx.close()


def test_spaceship_traversal_equals_oracle(pkg, oracle, spaceship):
    rng = np.random.default_rng(7)
    s = spaceship.scene
    lo, hi = np.array(s.bb_min[:]), np.array(s.bb_max[:])
    n = 20000
    start = lo + (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ctx = pkg.Context(0)
    ctx.upload_image(spaceship)
    t, surf, uv = ctx.intersect(start, d)
    t0, s0, uv0, cnt = oracle.intersect(spaceship, start, d)
    np.testing.assert_array_equal(t, t0)
    same = surf == s0
    assert (~same).sum() <= 5  # exact-t ties only
    np.testing.assert_array_equal(uv[same], uv0[same])
    assert (surf != 0xFFFFFFFF).sum() > n // 10
    ctx.close()
