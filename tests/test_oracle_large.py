"""The oracle against the reference on BASELINE configs[2..4] at full size (C3 metal_bunnies
quaternary SAH 1920x1080 @ 1024 spp, C4 spaceship 3840x2160 @ 1024 spp, C5 water_caustics photon map), the
missing meshes replaced by the deterministic stand-ins of integration/large_scenes/: a few full-width rows of each frame,
rendered by the reference itself at full resolution and spp (committed goldens), must come out of the oracle
bit for bit. Runs wherever oracle/_ref holds the reference binary and the scene copies (build container, GPU
box); about a minute on 8 cores plus, the first time, the flattening of the scenes."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_error

sys.path.insert(0, os.path.join(ROOT, "integration", "large_scenes"))


def test_synthetic_bunny_is_reproducible(tmp_path):
    import make_large
    import make_synthetic
    p = str(tmp_path / "bunny.obj")
    assert make_synthetic.write_bunny(p) == (40962, 81920)
    assert hashlib.md5(open(p, "rb").read()).hexdigest() == make_large.BUNNY_MD5


@pytest.mark.parametrize("name", ["c3", "c4", "c5", "c5_s16", "baroque", "lego", "pipes"])
def test_oracle_rows_equal_reference(pkg, oracle, name):
    import make_large
    p = make_large.ensure_image(name)  # also verifies the md5 of the generated meshes
    if p is None:
        pytest.skip("oracle/_ref (reference binary + scene copies) not on this machine")
    c = make_large.CONFIGS[name]
    img = pkg.SceneImage(p)
    assert (img.scene.num_surfaces, img.scene.num_nodes) == (c["surfaces"], c["nodes"])
    cam = img.camera
    if "base" in c:  # variant of a config: the base image with another spp (c5_s16: the 256 spp the bench times)
        cam.sqrtspp = c["sqrtspp"]
    assert (cam.width, cam.height, cam.sqrtspp) == (c["width"], c["height"], c["sqrtspp"])
    integ = pkg.INTEGRATOR_PHOTON_MAPPER if c["photon"] else pkg.INTEGRATOR_PATH_TRACER
    out, info = oracle.render(img, cam, make_large.SEED, integ, rows=c["rows"])
    ref = np.fromfile(make_large.golden_path(name)).reshape(out.shape)
    print("%s: %d paths, %.2f rays/path, %.1f box tests and %.1f primitive tests per ray" %
          (name, info["paths"], info["rays"] / info["paths"], info["node_tests"] / info["rays"], info["prim_tests"] / info["rays"]))
    if c["photon"]:
        # the map of this image was traced by another run of the reference than the golden rows: same photon set,
        # octree leaves filled in another (thread-dependent) order -> estimates summed in another order
        assert rel_error(out, ref).max() < 1e-9
    else:
        np.testing.assert_array_equal(out, ref)
