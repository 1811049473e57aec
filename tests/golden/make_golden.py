#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE ITSELF (oracle/_ref/mcrt_ref = the reference's
translation units compiled in place + oracle/ref_main.cpp). Runs only in the build container
(needs /root/reference); the outputs are committed so that the GPU box can check against them.

For every case it writes
  <name>.mcrt            scene image (flattened Scene/BVH/Camera[/photon maps]) — input of both the
                         oracle and the HIP library
  <name>.<tag>.f64       FP64 radiance camera.film.scan(x,y) of the reference for the listed rows
  <name>.<tag>.<save>.tga  Image::save of that frame (auto exposure/gain, tone map, sRGB bytes) per "saves" entry
  kat_<name>/            known-answer vectors of individual reference functions
and records everything in manifest.json. Seed: MCRT_REF_SEED (default 0x12345678).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
SCENES = "/root/reference/scenes"
SEED = 0x12345678

# name, scene json, extra args for the image, list of renders (tag, width, height, sqrtspp, rows or None)
CASES = [
    dict(name="hexagon_room_diffuse", scene="hexagon_room_diffuse.json", args=[],
         image=dict(width=256, height=256, sqrtspp=2),
         renders=[dict(tag="c1_256x256_s2", width=256, height=256, sqrtspp=2)], kat=2000),
    dict(name="hexagon_room", scene="hexagon_room.json", args=[],
         image=dict(width=1920, height=1080, sqrtspp=16),
         renders=[dict(tag="c2_192x108_s4", width=192, height=108, sqrtspp=4,
                       # Image::save with the scene's own "image" object (Hable, -0.25 EV) and with overrides of it
                       saves=[dict(tag="scene", opts="-"), dict(tag="aces_gain", opts="tonemapper=ACES,gain_compensation=-0.5"),
                              dict(tag="plain", opts="plain=true")]),
                  dict(tag="c2_1920x1080_s16_rows536_540", width=1920, height=1080, sqrtspp=16, rows=[536, 540])],
         kat=4000),
    dict(name="hexagon_room_ggx", scene="hexagon_room.json",
         args=["--specular-roughness", "green", "0.1", "--specular-roughness", "crystal", "0.05"],
         image=dict(width=1920, height=1080, sqrtspp=16),
         renders=[dict(tag="c2ggx_192x108_s4", width=192, height=108, sqrtspp=4),
                  # rows of the full-size frame bench.py's c2_ggx leg times ("report both")
                  dict(tag="c2ggx_1920x1080_s16_rows536_540", width=1920, height=1080, sqrtspp=16, rows=[536, 540])]),
    dict(name="hexagon_room_dof", scene="hexagon_room.json", args=["--f-stop", "1.8", "--focus-distance", "8"],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="dof_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="hexagon_room_pm", scene="hexagon_room.json", photon=True, args=["--emissions", "4000"],
         image=dict(width=96, height=72, sqrtspp=2),
         renders=[dict(tag="pm_96x72_s2", width=96, height=72, sqrtspp=2, saves=[dict(tag="scene", opts="-")])], kat=1000),
    dict(name="coffee_maker_qsah", scene="coffe_maker.json", args=["--bvh", "quaternary_sah"],
         image=dict(width=160, height=120, sqrtspp=2),
         renders=[dict(tag="cm_160x120_s2", width=160, height=120, sqrtspp=2)], kat=3000),
    dict(name="coffee_maker_bsah", scene="coffe_maker.json", args=["--bvh", "binary_sah"],
         image=dict(width=80, height=60, sqrtspp=2),
         renders=[dict(tag="cmb_80x60_s2", width=80, height=60, sqrtspp=2)]),
    dict(name="ior_test", scene="ior_test.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="ior_96x54_s3", width=96, height=54, sqrtspp=3)], kat=1000),
    dict(name="veach_mis", scene="veach_mis.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="veach_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="metals", scene="metals.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="metals_96x54_s3", width=96, height=54, sqrtspp=3, saves=[dict(tag="scene", opts="-")])]),
    dict(name="oren_nayar_test", scene="oren_nayar_test.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="on_96x54_s3", width=96, height=54, sqrtspp=3)]),
    # Surface::Quadric (no BASELINE config uses it): 13 sliced quadrics, octree BVH, thin lens, sky light
    dict(name="quadric", scene="quadric.json", args=[],
         image=dict(width=96, height=72, sqrtspp=3),
         renders=[dict(tag="quadric_96x72_s3", width=96, height=72, sqrtspp=3,
                       saves=[dict(tag="scene", opts="-"), dict(tag="bright", opts="exposure_compensation=2.5,gain_compensation=1")])], kat=3000),
    # Film reconstruction filters (no shipped scene has a "film" key): splats instead of per-pixel sums
    dict(name="film_mitchell", scene="hexagon_room.json", args=["--film-filter", "mitchell-netravali"],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="mitchell_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="film_gaussian_cached", scene="metals.json", args=["--film-filter", "gaussian", "--film-cache", "64"],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="gaussian_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="film_lanczos", scene="quadric.json", args=["--film-filter", "lanczos", "--film-radius", "1.5"],
         image=dict(width=96, height=72, sqrtspp=2),
         renders=[dict(tag="lanczos_96x72_s2", width=96, height=72, sqrtspp=2)]),
    # ... on the one shipped scene without a "bvh" key (Scene::intersect loops over the surfaces, scene.cpp:163-174)
    dict(name="film_gaussian_nobvh", scene="ior_test.json", args=["--film-filter", "gaussian"],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="gaussian_96x54_s3", width=96, height=54, sqrtspp=3)]),
    # ... the box filter with a radius other than its default 0.5 (film.cpp:44-46: still Filter::box = 1, over a wider window)
    dict(name="film_box_wide", scene="hexagon_room.json", args=["--film-filter", "box", "--film-radius", "1.3"],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="boxwide_96x54_s3", width=96, height=54, sqrtspp=3)]),
    # ... and on a photon-mapped frame
    dict(name="film_mitchell_pm", scene="hexagon_room.json", photon=True, args=["--emissions", "4000", "--film-filter", "mitchell-netravali"],
         image=dict(width=96, height=72, sqrtspp=2),
         renders=[dict(tag="mitchell_pm_96x72_s2", width=96, height=72, sqrtspp=2)]),
    # shell.json and stanford_dragon.json without the meshes of .MISSING_LARGE_BLOBS (the loader skips them): the rooms, their
    # 6 / 3 lights and 7 / 14 materials
    dict(name="shell_room", scene="shell.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="shell_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="dragon_room", scene="stanford_dragon.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="dragon_96x54_s3", width=96, height=54, sqrtspp=3)]),
    dict(name="ggx_test", scene="ggx_test.json", args=[],
         image=dict(width=96, height=54, sqrtspp=3),
         renders=[dict(tag="ggx_96x54_s3", width=96, height=54, sqrtspp=3)]),
]


def run(cmd):
    print("+", " ".join(cmd), flush=True)
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, env=dict(os.environ, MCRT_REF_SEED=str(SEED)))
    return out.stdout


def main():
    only = set(sys.argv[1:])
    manifest_path = os.path.join(HERE, "manifest.json")
    manifest = {"seed": SEED, "cases": {}}
    if only and os.path.exists(manifest_path):
        manifest = json.load(open(manifest_path))
    for case in CASES:
        if only and case["name"] not in only:
            continue
        base = [REF]
        common = ["--scene", os.path.join(SCENES, case["scene"])] + case["args"] + (["--photon"] if case.get("photon") else [])
        entry = dict(image=case["name"] + ".mcrt", photon=bool(case.get("photon")), renders=[])
        im = case["image"]
        first = True
        for r in case["renders"]:
            rad = "%s.%s.f64" % (case["name"], r["tag"])
            mode = "render"
            cmd = base + [mode] + common + ["--width", str(r["width"]), "--height", str(r["height"]), "--sqrtspp", str(r["sqrtspp"]),
                                            "--out-radiance", os.path.join(HERE, rad)]
            save_files = []
            for sv in r.get("saves", []):
                save_files.append("%s.%s.%s" % (case["name"], r["tag"], sv["tag"]))
                cmd += ["--save", os.path.join(HERE, save_files[-1]), sv["opts"]]
            rows = r.get("rows")
            if rows:
                cmd += ["--rows", str(rows[0]), str(rows[1])]
            same_as_image = (r["width"], r["height"], r["sqrtspp"]) == (im["width"], im["height"], im["sqrtspp"])
            # the image (and the KATs) come from the same process as the first render that has the
            # image's camera, so photon maps in the image are the ones the radiance was computed with
            if same_as_image and first:
                cmd[1] = "flatten,render" + (",kat" if case.get("kat") else "")
                cmd += ["--out", os.path.join(HERE, entry["image"])]
                if case.get("kat"):
                    cmd += ["--out-kat", os.path.join(HERE, "kat_" + case["name"]), "--n", str(case["kat"])]
                    entry["kat"] = "kat_" + case["name"]
                first = False
            out = run(cmd)
            info = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            entry["renders"].append(dict(file=rad, width=r["width"], height=r["height"], sqrtspp=r["sqrtspp"],
                                         rows=rows or [0, r["height"]], ref_seconds=info["seconds"], ref_threads=info["threads"]))
            if save_files:
                entry["renders"][-1]["saves"] = [dict(file=f + ".tga", **d) for f, d in zip(save_files, info["saves"])]
        if first:  # no render matched the image camera: flatten separately (path tracing only)
            assert not case.get("photon")
            cmd = base + ["flatten" + (",kat" if case.get("kat") else "")] + common + [
                "--width", str(im["width"]), "--height", str(im["height"]), "--sqrtspp", str(im["sqrtspp"]),
                "--out", os.path.join(HERE, entry["image"])]
            if case.get("kat"):
                cmd += ["--out-kat", os.path.join(HERE, "kat_" + case["name"]), "--n", str(case["kat"])]
                entry["kat"] = "kat_" + case["name"]
            run(cmd)
        # the function-level vectors that do not depend on the scene are kept once (hexagon_room)
        if entry.get("kat") and case["name"] != "hexagon_room":
            kd = os.path.join(HERE, entry["kat"])
            for f in os.listdir(kd):
                if f.startswith(("bsdf_", "sampler_")):
                    os.remove(os.path.join(kd, f))
        manifest["cases"][case["name"]] = entry
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)
    print("wrote", manifest_path)


if __name__ == "__main__":
    main()
