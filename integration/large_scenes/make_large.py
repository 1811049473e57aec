#!/usr/bin/env python3
"""Large-scene fixtures (too big for git). Everything here is produced by the REFERENCE (oracle/_ref/mcrt_ref
= the reference's translation units compiled in place + oracle/ref_main.cpp) under oracle/_ref/ (git-ignored),
from scene copies whose missing meshes (.MISSING_LARGE_BLOBS) are replaced by deterministic synthetic stand-ins
(integration/large_scenes/make_synthetic.py, integration/large_scenes/gen_mesh.c).

Build container (/root/reference present; main(), called from __graft_entry__.build()):
  oracle/_ref/bin/gen_mesh                              the C mesh generator
  oracle/_ref/scenes/{metal_bunnies,spaceship,water_caustics}.json + data/   scene copies with the meshes that exist
  oracle/_ref/images/spaceship.mcrt (+ .480x270_s2.f64)  spaceship.json as far as its meshes are present (68 760 tris)
  tests/golden/<config>.rows*.f64                         the reference's radiance for a few full-width rows of each
                                                        BASELINE config at full size — committed (<= 100 KB each)
Any machine that has oracle/_ref/ (build container and GPU box; ensure_image(name)):
  stand-in meshes + oracle/_ref/images/<config>.mcrt      flattened scene (120 MB - 1.7 GB: listed in .gpurunignore,
                                                        rebuilt on the GPU box in 5-60 s)
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
GEN = os.path.join(ROOT, "oracle", "_ref", "bin", "gen_mesh")
OUT = os.path.join(ROOT, "oracle", "_ref", "images")
SCENES_OUT = os.path.join(ROOT, "oracle", "_ref", "scenes")
SCENES = "/root/reference/scenes"
GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 0x12345678
BUNNY_MD5 = "3379d5cae7668b436c4a00c1a9e4bd74"

sys.path.insert(0, HERE)


def _bunny(path):
    import make_synthetic
    make_synthetic.write_bunny(path)


def _gen(*args):
    return lambda path: subprocess.check_call([GEN] + [str(a) for a in args] + [path])


# BASELINE configs[2..4] at full size. "meshes": generated file -> (generator, md5 of its bytes).
CONFIGS = {
    "c3": dict(scene="metal_bunnies.json", copy=["data/shelf.obj", "data/backwall.obj"],
               meshes={"data/bunny.obj": (_bunny, BUNNY_MD5)},
               flags=["--bvh", "quaternary_sah", "--bins", "8", "--width", "1920", "--height", "1080", "--sqrtspp", "32"],
               width=1920, height=1080, sqrtspp=32, rows=(540, 542), photon=False,
               golden="metal_bunnies_c3.rows540_542.f64", image="metal_bunnies_c3.mcrt",
               surfaces=491593, nodes=169162),
    "c4": dict(scene="spaceship.json", copy=["data/spaceship", "data/spectral-distributions"],
               meshes={"data/spaceship/aluminium.obj": (_gen("bowl", 500, 300, -0.385, 0.73, 0, 1.6, 1.0, 1.6, 0.65, -0.76, 21),
                                                        "313e7f20fc8507667801e81c30b1721c"),
                       "data/spaceship/steel.obj": (_gen("bowl", 220, 201, -0.385, 0.73, 0, 1.45, 0.9, 1.45, 0.65, -0.76, 22),
                                                    "7f8d2185b55dcc6c6c7fbdb591330556")},
               flags=["--width", "3840", "--height", "2160", "--sqrtspp", "32"],
               width=3840, height=2160, sqrtspp=32, rows=(1080, 1081), photon=False,
               golden="spaceship_c4.rows1080_1081.f64", image="spaceship_c4.mcrt",
               surfaces=457200, nodes=153801),
    # photon map of the parity image: 1e5 emissions x caustic_factor 10 traced by the reference; the timing run
    # (bench.py --workload c5) emits on the GPU
    "c5": dict(scene="water_caustics.json", copy=["data/water_caustics"],
               meshes={"data/water_caustics/water.obj": (_gen("water", 1835), "3728d20d4225dd8617842d74e25a4f6e"),
                       "data/bunny.obj": (_bunny, BUNNY_MD5)},
               flags=["--photon", "--emissions", "100000", "--width", "1000", "--height", "1000", "--sqrtspp", "4"],
               width=1000, height=1000, sqrtspp=4, rows=(500, 504), photon=True,
               golden="water_caustics_c5.rows500_504.f64", image="water_caustics_c5.mcrt",
               surfaces=6898815, nodes=1925901),
}
# Three more of the reference's own scenes, as far as their meshes are in the tree (.MISSING_LARGE_BLOBS lists the rest; the
# reference's loader skips a missing file): real meshes, metals with tabulated spectra, glass, 2 - 546 emissive surfaces, the
# scene files' own cameras (16 spp), quaternary-SAH trees of 18 k - 124 k nodes. No stand-in meshes.
CONFIGS["baroque"] = dict(scene="baroque_table.json", copy=["data/baroque_table", "data/spectral-distributions"], meshes={}, flags=[],
                          width=1280, height=720, sqrtspp=4, rows=(360, 363), photon=False,
                          golden="baroque_table.rows360_363.f64", image="baroque_table.mcrt", surfaces=51304, nodes=18249)
CONFIGS["lego"] = dict(scene="lego_bulldozer.json", copy=["data/lego_bulldozer"], meshes={}, flags=[],
                       width=1280, height=720, sqrtspp=4, rows=(360, 363), photon=False,
                       golden="lego_bulldozer.rows360_363.f64", image="lego_bulldozer.mcrt", surfaces=122917, nodes=41304)
CONFIGS["pipes"] = dict(scene="pipes.json", copy=["data/pipes", "data/spectral-distributions"], meshes={}, flags=[],
                        width=960, height=600, sqrtspp=4, rows=(300, 304), photon=False,
                        golden="pipes.rows300_304.f64", image="pipes.mcrt", surfaces=357765, nodes=124259)
# The same C5 scene and photon map at the spp the bench times (256 instead of the scene file's 16): one full-width row by the
# reference (its photon pass is reproducible: same photon sets for the same seed, radiance equal to 1e-15). Rendered against the
# c5 image with the camera's sqrtspp overridden (tests/test_gpu_large_scene.py).
CONFIGS["c5_s16"] = dict(CONFIGS["c5"], flags=["--photon", "--emissions", "100000", "--width", "1000", "--height", "1000", "--sqrtspp", "16"],
                         sqrtspp=16, rows=(500, 501), golden="water_caustics_c5.s16_rows500_501.f64", separate_render=True, base="c5")
C3 = dict(CONFIGS["c3"], golden=os.path.join(GOLDEN, CONFIGS["c3"]["golden"]), image=os.path.join(OUT, CONFIGS["c3"]["image"]))


def _run(cmd):
    subprocess.check_call(cmd, env=dict(os.environ, MCRT_REF_SEED=str(SEED)), stdout=subprocess.DEVNULL)


def golden_path(name):
    return os.path.join(GOLDEN, CONFIGS[name]["golden"])


def image_path(name):
    return os.path.join(OUT, CONFIGS[name]["image"])


def spaceship(force=False):
    os.makedirs(OUT, exist_ok=True)
    img = os.path.join(OUT, "spaceship.mcrt")
    rad = os.path.join(OUT, "spaceship.480x270_s2.f64")
    if force or not (os.path.exists(img) and os.path.exists(rad)):
        _run([REF, "flatten,render", "--scene", os.path.join(SCENES, "spaceship.json"), "--width", "480", "--height", "270",
              "--sqrtspp", "2", "--out", img, "--out-radiance", rad])
    return img, rad


def build_generator():
    os.makedirs(os.path.dirname(GEN), exist_ok=True)
    src = os.path.join(HERE, "gen_mesh.c")
    if not os.path.exists(GEN) or os.path.getmtime(src) > os.path.getmtime(GEN):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", GEN, src, "-lm"])


def prepare_scene(name):
    """Scene copy with the meshes the reference tree does have (needs /root/reference)."""
    c = CONFIGS[name]
    os.makedirs(os.path.join(SCENES_OUT, "data"), exist_ok=True)
    shutil.copy(os.path.join(SCENES, c["scene"]), os.path.join(SCENES_OUT, c["scene"]))
    for rel in c["copy"]:
        src, dst = os.path.join(SCENES, rel), os.path.join(SCENES_OUT, rel)
        if os.path.isdir(src):
            shutil.copytree(src, dst, dirs_exist_ok=True)
        else:
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copy(src, dst)
    if name == "c4":
        # spaceship.json as the reference tree has it (the two hull meshes of .MISSING_LARGE_BLOBS absent, 68 760 triangles):
        # the scene behind oracle/_ref/images/spaceship.mcrt, for the reference leg of bench.py's "spaceship" workload
        text = open(os.path.join(SCENES, c["scene"])).read()
        for mesh in ("aluminium.obj", "steel.obj"):
            assert "data/spaceship/" + mesh in text
            text = text.replace("data/spaceship/" + mesh, "data/spaceship/absent-" + mesh)
        with open(os.path.join(SCENES_OUT, "spaceship_cockpit.json"), "w") as f:
            f.write(text)
    for root, dirs, files in os.walk(SCENES_OUT):  # the reference tree is read-only; the copies must not be
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
        for d in dirs:
            os.chmod(os.path.join(root, d), 0o755)


def ensure_meshes(name):
    """Writes the stand-in meshes of a config if missing and checks their fingerprints."""
    c = CONFIGS[name]
    if not os.path.exists(os.path.join(SCENES_OUT, c["scene"])):
        return False
    for rel, (gen, md5) in c["meshes"].items():
        path = os.path.join(SCENES_OUT, rel)
        if not os.path.exists(path):
            if gen is not _bunny and not os.path.exists(GEN):
                return False
            os.makedirs(os.path.dirname(path), exist_ok=True)
            gen(path)
        h = hashlib.md5()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        if h.hexdigest() != md5:
            raise RuntimeError("%s differs from the committed fingerprint: %s" % (rel, h.hexdigest()))
    return True


def scene_flags(name):
    c = CONFIGS[name]
    return ["--scene", os.path.join(SCENES_OUT, c["scene"])] + c["flags"]


def make_golden(name, force=False):
    """A few full-width rows of the config's frame rendered by the reference at full resolution and spp."""
    c = CONFIGS[name]
    g = golden_path(name)
    if force or not os.path.exists(g):
        ensure_meshes(name)
        if c.get("separate_render"):  # rows only; the image is the base config's
            _run([REF, "render"] + scene_flags(name) + ["--rows", str(c["rows"][0]), str(c["rows"][1]), "--out-radiance", g])
        elif c["photon"]:
            # the rows and the image must come from the same process (one photon map): keep the image too
            os.makedirs(OUT, exist_ok=True)
            _run([REF, "flatten,render"] + scene_flags(name) + ["--rows", str(c["rows"][0]), str(c["rows"][1]), "--out", image_path(name),
                                                                "--out-radiance", g])
        else:
            _run([REF, "render"] + scene_flags(name) + ["--rows", str(c["rows"][0]), str(c["rows"][1]), "--out-radiance", g])
    return g


def ensure_image(name):
    """Flatten the config's scene with the reference's loader and BVH builder (and, for c5, its photon pass).
    Returns the path, or None when the reference binary / scene copy is not on this machine."""
    name = CONFIGS[name].get("base", name)  # variants share the base config's image
    p = image_path(name)
    if os.path.exists(p):
        return p
    if not os.path.exists(REF) or not ensure_meshes(name):
        return None
    os.makedirs(OUT, exist_ok=True)
    _run([REF, "flatten"] + scene_flags(name) + ["--out", p])
    return p


def ensure_c3_image():
    return ensure_image("c3")


def main(force=False):
    build_generator()
    spaceship(force)
    for name in CONFIGS:
        if "base" not in CONFIGS[name]:
            prepare_scene(name)
        make_golden(name, force)


if __name__ == "__main__":
    main(force="--force" in sys.argv)
    for n in sys.argv[1:]:
        if n in CONFIGS:
            print(ensure_image(n))
