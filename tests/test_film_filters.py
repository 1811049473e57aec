"""Film reconstruction filters (camera/film.cpp:19-113, camera/filter.hpp). The goldens were
rendered by the reference with a "film" key injected into the camera (oracle/ref_main.cpp --film-filter …): every sample
is a splat over the pixels within the filter radius, accumulated with atomics — the reference's own sums therefore
depend on its thread timing in the last bits, and so do everyone else's: tolerance 1e-12, not bit equality."""
import ctypes as C

import numpy as np
import pytest

from conftest import camera_for, golden_path, load_radiance, rel_error

CASES = ["film_mitchell", "film_gaussian_cached", "film_lanczos"]
TOL = 1e-12


def _case(pkg, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    r = case["renders"][0]
    return img, camera_for(img, r), r


@pytest.mark.parametrize("name", CASES)
def test_oracle_film_equals_reference(pkg, oracle, manifest, name):
    img, cam, r = _case(pkg, manifest, name)
    assert cam.film_filter != 0 and cam.film_radius > 0
    out, _ = oracle.render(img, cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert rel_error(out, load_radiance(r)).max() < TOL


def test_filter_is_not_a_no_op(pkg, oracle, manifest):
    img, cam, r = _case(pkg, manifest, "film_mitchell")
    box = cam.copy()
    box.film_filter, box.film_radius, box.film_cache_size = 0, 0.0, 0
    out, _ = oracle.render(img, box, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert rel_error(out, load_radiance(r)).max() > 1e-3


@pytest.mark.parametrize("name,slots", [("film_mitchell", 777), ("film_gaussian_cached", 100000), ("film_lanczos", 64)])
def test_wavefront_device_code_film(pkg, emu, manifest, name, slots):
    """mcrt_film.hpp + the splat branch of wfShadeSlot, host build."""
    img, cam, r = _case(pkg, manifest, name)
    out = np.zeros((cam.height, cam.width, 3))
    cnt = (C.c_uint64 * 6)()
    assert emu.emu_render_wf(C.byref(img.scene), C.byref(cam), manifest["seed"], slots, cam.height, out.ctypes.data, cnt) == 0
    assert rel_error(out, load_radiance(r)).max() < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_film_matches_reference(pkg, manifest, name):
    img, cam, r = _case(pkg, manifest, name)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    rel = rel_error(out, load_radiance(r)).max(axis=2)
    print("%s: max rel %.3e" % (name, rel.max()))
    assert (rel > 1e-4).sum() <= max(2, int(0.002 * rel.size)) and np.quantile(rel, 0.99) < 1e-9
    assert st["kernel_launches"] > 2  # the wavefront pipeline
    ctx.close()


@pytest.mark.gpu
def test_gpu_film_unsupported_combinations(pkg, manifest):
    img, cam, _ = _case(pkg, manifest, "film_mitchell")
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    sharded = cam.copy()
    sharded.shard_count, sharded.shard_index, sharded.shard_rows = 2, 0, 8
    with pytest.raises(pkg.McrtError) as e:
        ctx.sample_image(sharded, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert "unsharded" in str(e.value)
    ctx.close()
