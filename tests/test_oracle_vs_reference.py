"""Pins the oracle (oracle/mcrt_oracle.c) against outputs of the reference itself: the FP64 radiance
dumps and function-level known-answer vectors under tests/golden/ were produced by the reference's
own translation units (tests/golden/make_golden.py -> oracle/_ref/mcrt_ref). The reference ships no
tests or golden vectors, so these dumps are the anchor. CPU only."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import camera_for, golden_path, load_radiance, rel_error


def _cases(manifest):
    return [(n, c, r) for n, c in manifest["cases"].items() for r in c["renders"]]


def test_manifest_has_all_configs(manifest):
    assert {"hexagon_room_diffuse", "hexagon_room", "hexagon_room_ggx", "hexagon_room_pm"} <= set(manifest["cases"])


def test_c1_anchor_matches_survey_appendix_b2():
    # sha256 of the C1 dump produced during the survey from the reference
    h = hashlib.sha256(open(golden_path("hexagon_room_diffuse.c1_256x256_s2.f64"), "rb").read()).hexdigest()
    assert h == "2d1014101b957ce88e988540e45f74d226992c5c57e1a188dd2577b50f0f7c26"


@pytest.mark.parametrize("name", ["hexagon_room_diffuse", "hexagon_room", "hexagon_room_ggx", "hexagon_room_pm", "hexagon_room_dof",
                                  "coffee_maker_qsah", "coffee_maker_bsah", "ior_test", "veach_mis", "metals",
                                  "oren_nayar_test", "ggx_test", "quadric", "shell_room", "dragon_room"])
def test_oracle_radiance_equals_reference(pkg, oracle, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    integ = pkg.INTEGRATOR_PHOTON_MAPPER if case["photon"] else pkg.INTEGRATOR_PATH_TRACER
    for r in case["renders"]:
        if r["width"] * (r["rows"][1] - r["rows"][0]) * r["sqrtspp"] ** 2 > 3_000_000:
            continue  # the full-resolution crop is checked in test_oracle_c2_full_resolution_rows
        ref = load_radiance(r)
        out, info = oracle.render(img, camera_for(img, r), manifest["seed"], integ, rows=r["rows"])
        # same libm, same operation order -> identical bits (the golden files were made in this image)
        assert np.array_equal(out, ref), "max rel err %.3e" % rel_error(out, ref).max()
        assert info["paths"] == ref.shape[0] * ref.shape[1] * r["sqrtspp"] ** 2


def test_oracle_c2_full_resolution_rows(pkg, oracle, manifest):
    case = manifest["cases"]["hexagon_room"]
    r = [x for x in case["renders"] if x["width"] == 1920][0]
    img = pkg.SceneImage(golden_path(case["image"]))
    out, _ = oracle.render(img, camera_for(img, r), manifest["seed"], pkg.INTEGRATOR_PATH_TRACER, rows=r["rows"])
    assert np.array_equal(out, load_radiance(r))


def test_sampler_known_answers_survey_b1(oracle):
    # SURVEY.md appendix B.1 (values printed by the reference's Sampler with global_seed 0x12345678)
    seed = 0x12345678
    np.testing.assert_array_equal(oracle.sampler(seed, 0, 0, 0)[:4],
                                  [0.91819469165056944, 0.18568774103187025, 0.79369318997487426, 0.5791851484682411])
    np.testing.assert_array_equal(oracle.sampler(seed, 0, 0, 1),
                                  [0.55389368417672813, 0.72998381359502673, 0.35659338510595262, 0.088447618996724486,
                                   0.34757957025431097, 0.65755581227131188, 0.59028820460662246])
    np.testing.assert_array_equal(oracle.sampler(seed, 0, 0, 2),
                                  [0.068414845271036029, 0.84792701038531959, 0.37911035283468664, 0.64697259897366166,
                                   0.88456402625888586, 0.97397215408273041, 0.98102473746985197])
    np.testing.assert_array_equal(oracle.sampler(seed, 2073599, 1, 1)[:3],
                                  [0.55678080138750374, 0.66295242263004184, 0.79433509847149253])


def test_sampler_kat(oracle, manifest):
    d = golden_path(manifest["cases"]["hexagon_room"]["kat"])
    inp = np.fromfile(os.path.join(d, "sampler_in.u32"), dtype=np.uint32).reshape(-1, 3)
    ref = np.fromfile(os.path.join(d, "sampler_out.f64")).reshape(-1, 7)
    for (pixel, index, shuffles), want in zip(inp, ref):
        np.testing.assert_array_equal(oracle.sampler(manifest["seed"], pixel, index, shuffles), want)


@pytest.mark.parametrize("name", ["hexagon_room", "hexagon_room_diffuse", "coffee_maker_qsah", "ior_test", "quadric"])
def test_intersect_kat(pkg, oracle, manifest, name):
    case = manifest["cases"][name]
    img = pkg.SceneImage(golden_path(case["image"]))
    d = golden_path(case["kat"])
    rays = np.fromfile(os.path.join(d, "isect_rays.f64")).reshape(-1, 6)
    t, surf, uv, _ = oracle.intersect(img, rays[:, :3].copy(), rays[:, 3:].copy())
    np.testing.assert_array_equal(surf, np.fromfile(os.path.join(d, "isect_surface.u32"), dtype=np.uint32))
    np.testing.assert_array_equal(t, np.fromfile(os.path.join(d, "isect_t.f64")))
    np.testing.assert_array_equal(uv, np.fromfile(os.path.join(d, "isect_uv.f64")).reshape(-1, 2))
    assert (surf != 0xFFFFFFFF).sum() > len(surf) // 20  # the vectors exercise hits as well as misses (quadric.json is sparse)


def test_bsdf_kat(oracle, manifest):
    d = golden_path(manifest["cases"]["hexagon_room"]["kat"])
    inp = np.fromfile(os.path.join(d, "bsdf_in.f64")).reshape(-1, 11)
    ref = np.fromfile(os.path.join(d, "bsdf_out.f64")).reshape(-1, 18)
    consts = np.fromfile(os.path.join(d, "bsdf_consts.f64"))
    out = oracle.bsdf_kat(inp, consts)
    np.testing.assert_array_equal(out, ref)


def test_knn_kat(pkg, oracle, manifest):
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    k = img.param("k_nearest_photons")
    d = golden_path(case["kat"])
    for which, tag in ((0, "g"), (1, "c")):
        pts = np.fromfile(os.path.join(d, "knn_%s_points.f64" % tag)).reshape(-1, 3)
        cnt, idx, d2 = oracle.knn(img.photons(which), pts, k)
        np.testing.assert_array_equal(cnt, np.fromfile(os.path.join(d, "knn_%s_count.u32" % tag), dtype=np.uint32))
        np.testing.assert_array_equal(d2, np.fromfile(os.path.join(d, "knn_%s_d2.f64" % tag)).reshape(-1, k))
        np.testing.assert_array_equal(idx, np.fromfile(os.path.join(d, "knn_%s_index.u32" % tag), dtype=np.uint32).reshape(-1, k))
