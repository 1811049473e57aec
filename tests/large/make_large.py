#!/usr/bin/env python3
"""Large-scene fixtures (too big for git): made by the REFERENCE in the build container into
oracle/_ref/images/ (git-ignored, travels to the GPU box with the snapshot like every other file under
oracle/_ref/). Called from __graft_entry__.build() when /root/reference is present.

  spaceship.mcrt              spaceship.json (BASELINE configs[3] scene; 68 760 of its 457 200 triangles
                              are present in the reference tree, see .MISSING_LARGE_BLOBS), quaternary SAH
                              BVH built by the reference: 23 187 nodes, 363 materials, 354 emissive triangles
  spaceship.480x270_s2.f64    the reference's FP64 radiance, 480x270 @ 4 spp
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref")
OUT = os.path.join(ROOT, "oracle", "_ref", "images")
SCENES = "/root/reference/scenes"


def main(force=False):
    os.makedirs(OUT, exist_ok=True)
    img = os.path.join(OUT, "spaceship.mcrt")
    rad = os.path.join(OUT, "spaceship.480x270_s2.f64")
    if force or not (os.path.exists(img) and os.path.exists(rad)):
        subprocess.check_call([REF, "flatten,render", "--scene", os.path.join(SCENES, "spaceship.json"), "--width", "480",
                               "--height", "270", "--sqrtspp", "2", "--out", img, "--out-radiance", rad],
                              env=dict(os.environ, MCRT_REF_SEED=str(0x12345678)), stdout=subprocess.DEVNULL)
    return img, rad


if __name__ == "__main__":
    print(main(force="--force" in sys.argv))
