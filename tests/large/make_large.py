#!/usr/bin/env python3
"""Large-scene fixtures (too big for git). Everything here is produced by the REFERENCE (oracle/_ref/mcrt_ref
= the reference's translation units compiled in place + oracle/ref_main.cpp) under oracle/_ref/ (git-ignored).

Build container (/root/reference present; called from __graft_entry__.build()):
  oracle/_ref/images/spaceship.mcrt (+ .480x270_s2.f64)   spaceship.json as far as its meshes are present
                              (68 760 of 457 200 triangles, .MISSING_LARGE_BLOBS), quaternary SAH, 23 187 nodes
  oracle/_ref/scenes/metal_bunnies.json + data/            scene copy for BASELINE configs[2] with the synthetic
                              stand-in bunny.obj (tests/large/make_synthetic.py), shelf.obj, backwall.obj
  tests/golden/metal_bunnies_c3.rows540_542.f64             the reference's radiance for two full-width rows of
                              the C3 frame (1920x1080 @ 1024 spp, quaternary SAH) — committed (92 KB)
Any machine that has oracle/_ref/ (build container and GPU box):
  oracle/_ref/images/metal_bunnies_c3.mcrt                  flattened C3 scene, 491 592 primitives (about 120 MB:
                              listed in .gpurunignore, rebuilt on the GPU box by ensure_c3_image() in ~10 s)
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
OUT = os.path.join(ROOT, "oracle", "_ref", "images")
SCENES_OUT = os.path.join(ROOT, "oracle", "_ref", "scenes")
SCENES = "/root/reference/scenes"
SEED = 0x12345678
BUNNY_MD5 = "3379d5cae7668b436c4a00c1a9e4bd74"
C3 = dict(width=1920, height=1080, sqrtspp=32, rows=(540, 542), bvh="quaternary_sah", bins=8,
          golden=os.path.join(ROOT, "tests", "golden", "metal_bunnies_c3.rows540_542.f64"),
          image=os.path.join(OUT, "metal_bunnies_c3.mcrt"))

sys.path.insert(0, HERE)


def _run(cmd):
    subprocess.check_call(cmd, env=dict(os.environ, MCRT_REF_SEED=str(SEED)), stdout=subprocess.DEVNULL)


def spaceship(force=False):
    os.makedirs(OUT, exist_ok=True)
    img = os.path.join(OUT, "spaceship.mcrt")
    rad = os.path.join(OUT, "spaceship.480x270_s2.f64")
    if force or not (os.path.exists(img) and os.path.exists(rad)):
        _run([REF, "flatten,render", "--scene", os.path.join(SCENES, "spaceship.json"), "--width", "480", "--height", "270",
              "--sqrtspp", "2", "--out", img, "--out-radiance", rad])
    return img, rad


def prepare_c3_scene(force=False):
    """Scene directory for metal_bunnies with the synthetic bunny (needs /root/reference)."""
    import make_synthetic
    data = os.path.join(SCENES_OUT, "data")
    os.makedirs(data, exist_ok=True)
    shutil.copy(os.path.join(SCENES, "metal_bunnies.json"), os.path.join(SCENES_OUT, "metal_bunnies.json"))
    for f in ("shelf.obj", "backwall.obj"):
        shutil.copy(os.path.join(SCENES, "data", f), os.path.join(data, f))
    bunny = os.path.join(data, "bunny.obj")
    if force or not os.path.exists(bunny):
        make_synthetic.write_bunny(bunny)
    md5 = hashlib.md5(open(bunny, "rb").read()).hexdigest()
    if md5 != BUNNY_MD5:
        raise RuntimeError("synthetic bunny.obj differs from the committed fingerprint: %s" % md5)


def c3_flags():
    return ["--scene", os.path.join(SCENES_OUT, "metal_bunnies.json"), "--bvh", C3["bvh"], "--bins", str(C3["bins"]),
            "--width", str(C3["width"]), "--height", str(C3["height"]), "--sqrtspp", str(C3["sqrtspp"])]


def c3_golden(force=False):
    """Two full-width rows of the C3 frame rendered by the reference (about 4 M paths)."""
    if force or not os.path.exists(C3["golden"]):
        _run([REF, "render"] + c3_flags() + ["--rows", str(C3["rows"][0]), str(C3["rows"][1]), "--out-radiance", C3["golden"]])
    return C3["golden"]


def ensure_c3_image():
    """Flatten the C3 scene with the reference's loader and BVH builder. Returns the path, or None when
    the reference binary / scene copy is not on this machine."""
    if os.path.exists(C3["image"]):
        return C3["image"]
    if not (os.path.exists(REF) and os.path.exists(os.path.join(SCENES_OUT, "metal_bunnies.json"))
            and os.path.exists(os.path.join(SCENES_OUT, "data", "bunny.obj"))):
        return None
    os.makedirs(OUT, exist_ok=True)
    _run([REF, "flatten"] + c3_flags() + ["--out", C3["image"]])
    return C3["image"]


def main(force=False):
    spaceship(force)
    prepare_c3_scene(force)
    c3_golden(force)


if __name__ == "__main__":
    main(force="--force" in sys.argv)
    print(ensure_c3_image())
