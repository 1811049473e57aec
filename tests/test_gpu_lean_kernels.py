"""Lean kernel instances (csrc/mcrt_hip_lean.hip): the default path's kernels compiled WITHOUT the material features a scene does not
use - rough diffuse (Oren-Nayar, material/material.cpp:17-27), rough specular (GGX, material/ggx.cpp:21-88) and conductor Fresnel
(material/fresnel.cpp:16-49). launchRender picks them for a scene none of whose materials carries MCRT_MAT_ROUGH / ROUGH_SPECULAR /
COMPLEX_IOR; MCRT_LEAN_KERNELS=0 keeps the full instances. Same source, same arithmetic on every path such a scene can take: the frames
must be the full kernels' BITS - flat megakernel, state machine, photon-mapping kernel, pipeline, emission - and a scene that does use
one of the features must never get a lean instance."""
import os

import numpy as np
import pytest

from conftest import golden_path, camera_for

pytestmark = pytest.mark.gpu

ROUGH_BITS = 1 | 2 | 64  # MCRT_MAT_ROUGH | MCRT_MAT_ROUGH_SPECULAR | MCRT_MAT_COMPLEX_IOR (include/mcrt.h)


def _uses_rough(img):
    s = img.scene
    return any(s.materials[i].flags & ROUGH_BITS for i in range(s.num_materials))


@pytest.fixture
def env():
    keys = ("MCRT_KERNEL", "MCRT_LEAN_KERNELS", "MCRT_WF_PM_MIN_PATHS")
    old = {k: os.environ.get(k) for k in keys}
    yield os.environ
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("name,kernel", [("hexagon_room", None), ("hexagon_room_dof", None), ("hexagon_room_diffuse", None), ("veach_mis", None),
                                         ("ior_test", None), ("coffee_maker_qsah", None), ("coffee_maker_qsah", "wf"), ("hexagon_room", "wf"),
                                         ("dragon_room", None), ("shell_room", "wf")])
def test_lean_instances_render_the_full_instances_bits(pkg, manifest, env, name, kernel):
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    img = pkg.SceneImage(golden_path(case["image"]))
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])
    if kernel:
        env["MCRT_KERNEL"] = kernel
    env["MCRT_LEAN_KERNELS"] = "0"
    full, st0 = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert ctx.get_option("MCRT_LEAN_USED") == "0"
    env.pop("MCRT_LEAN_KERNELS")
    out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    lean = ctx.get_option("MCRT_LEAN_USED") == "1"
    assert lean == (not _uses_rough(img)), "%s: lean instance %s, materials rough: %s" % (name, lean, _uses_rough(img))
    assert st["kernel_id"] == st0["kernel_id"] and st["rays"] == st0["rays"]
    np.testing.assert_array_equal(out, full)
    print("%s (%s): lean instance %s, %d rays, same bits" % (name, pkg.KERNEL_NAMES.get(st["kernel_id"]), "USED" if lean else "not used (rough materials)", st["rays"]))
    ctx.close()


@pytest.mark.parametrize("name", ["hexagon_room_ggx", "metals", "ggx_test", "oren_nayar_test"])
def test_scenes_with_rough_materials_keep_the_full_instances(pkg, manifest, name):
    case = manifest["cases"].get(name)
    if case is None:
        pytest.skip("no such golden case")
    img = pkg.SceneImage(golden_path(case["image"]))
    assert _uses_rough(img)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    cam = camera_for(img, case["renders"][0])
    ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PATH_TRACER)
    assert ctx.get_option("MCRT_LEAN_USED") == "0"
    ctx.close()


@pytest.mark.parametrize("kernel", [None, "wf"])
def test_photon_mapped_frame_and_photon_pass_through_lean_instances(pkg, manifest, env, kernel):
    """hexagon_room_pm: the photon-mapping megakernel (or the pipeline's shade launch) and the emission kernel as lean instances - the
    frame and the emitted photon lists are the full instances' bits."""
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    assert not _uses_rough(img)
    cam = camera_for(img, case["renders"][0])
    frames, lists = [], []
    for lean in (False, True):
        if lean:
            env.pop("MCRT_LEAN_KERNELS", None)
        else:
            env["MCRT_LEAN_KERNELS"] = "0"
        if kernel:
            env["MCRT_KERNEL"] = kernel
        ctx = pkg.Context(0)
        ctx.upload_image(img)
        out, st = ctx.sample_image(cam, manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
        assert (ctx.get_option("MCRT_LEAN_USED") == "1") == lean
        frames.append(out)
        e = ctx.emit_photons(2.0e5, 10.0, manifest["seed"])
        assert (ctx.get_option("MCRT_LEAN_USED") == "1") == lean
        by_key = lambda ph, keys: ph[np.argsort(keys, kind="stable")]  # (a list's order is the waves' timing; its (light, emission, bounce) keys are not)
        lists.append((by_key(*e["global_"]), by_key(*e["caustic"])))
        assert len(e["global_"][1]) > 1000 and len(e["caustic"][1]) > 100
        ctx.close()
    np.testing.assert_array_equal(frames[1], frames[0])
    np.testing.assert_array_equal(lists[1][0], lists[0][0])
    np.testing.assert_array_equal(lists[1][1], lists[0][1])


def test_large_photon_mapped_frame_of_a_tree_in_memory_goes_through_the_pipeline(pkg, env):
    """launchRender's rule (round 6): a photon-mapped frame of a scene whose tree stays in memory and whose materials allow the lean
    instances goes through the pipeline - lean shade launch, trace launch, lean kNN launch at 6 waves per SIMD - once it has
    MCRT_WF_PM_MIN_PATHS path samples (default 32 M); below that, and with the option out of reach, the megakernel renders it. The two
    are the same bits (every estimate is the same wave-cooperative search and sum)."""
    from test_gpu_large_scene import _config
    img, c, _ = _config(pkg, "c5")
    assert not _uses_rough(img)
    ctx = pkg.Context(0)
    ctx.upload_image(img)
    ctx.upload_photons(img.photons(0), img.photons(1), img.param("k_nearest_photons"), bool(img.param("direct_visualization")))
    cam = img.camera
    cam.sqrtspp = 2
    r0, r1 = 496, 504
    cam.shard_rows, cam.shard_count = r1 - r0, (cam.height + r1 - r0 - 1) // (r1 - r0)
    cam.shard_index = r0 // (r1 - r0)
    mega, st0 = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PHOTON_MAPPER)
    assert st0["kernel_id"] == pkg.KERNEL_PM_WAVE  # 32 000 path samples: far below the rule
    env["MCRT_WF_PM_MIN_PATHS"] = "1000"
    try:
        pipe, st1 = ctx.sample_image(cam, 0x12345678, pkg.INTEGRATOR_PHOTON_MAPPER)
    finally:
        env.pop("MCRT_WF_PM_MIN_PATHS")
    assert st1["kernel_id"] == pkg.KERNEL_WAVEFRONT_PM and ctx.get_option("MCRT_LEAN_USED") == "1"
    assert st1["knn_searches"] == st0["knn_searches"] > 0
    np.testing.assert_array_equal(pipe, mega)
    ctx.close()
