"""The drop-in, run for real: oracle/_ref/mcrt_ref_gpu is the reference's OWN main() and all of its translation units, compiled
unmodified (oracle/Makefile, target `dropin`), with ONE function replaced — Camera::sampleImage() (camera/camera.cpp:101-145) —
by integration/camera_sample_image_gpu.cpp, which flattens the reference's Scene / BVH / photon maps / Camera and renders
through libmcrt_hip.so. The binary is driven the way a user drives the reference: a directory of scene files as argv, the menu
answers on stdin; Camera::capture() then calls the GPU sampleImage and the reference's own Image::save writes `<savename>.tga`.

Checked: the .tga against the one the unmodified reference wrote for the same scene, size and seed (tests/golden, made by
tests/golden/make_golden.py), and the FP64 frame handed to camera.image against the reference's radiance (1e-4, as everywhere)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_path, rel_error

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "oracle", "_ref", "mcrt_ref_gpu")
SCENES = os.path.join(ROOT, "oracle", "_ref", "scenes")


def _scene_dir(tmp_path, scene, render, emissions=None):
    """A scene directory as the reference expects it (scenes/*.json + data/), holding ONE scene whose camera 0 has the golden
    render's image size and spp — the only keys changed (the reference has no command-line overrides)."""
    d = tmp_path / "scenes"
    d.mkdir()
    j = json.load(open(os.path.join(SCENES, scene)))
    cam = j["cameras"][0]
    cam["image"]["width"], cam["image"]["height"], cam["sqrtspp"] = render["width"], render["height"], render["sqrtspp"]
    j["cameras"] = [cam]
    if emissions:
        j["photon_map"]["emissions"] = emissions
    with open(d / scene, "w") as f:
        json.dump(j, f)
    if os.path.isdir(os.path.join(SCENES, "data")):
        os.symlink(os.path.join(SCENES, "data"), d / "data")
    return d, cam["savename"]


# extra: the drop-in's switches (integration/camera_sample_image_gpu.cpp). Default = one context per device the process sees and the
# photon pass on the GPU (the reference's PhotonMapper constructor replaced); MCRT_DROPIN_CONTEXTS=2/3: the fan-out over several
# contexts (mcrt_render_multi, and since round 6 the SHARDED photon pass mcrt_photon_pass_multi) exercised on the one GPU of the test box;
# MCRT_DROPIN_REPLICATED_PHOTONS=1: round 5's photon pass (every context traces all paths); MCRT_DROPIN_CPU_PHOTONS=1: the reference's own CPU photon pass.
@pytest.mark.parametrize("name,scene,answers,emissions,extra", [
    ("hexagon_room", "hexagon_room.json", "0\nn\n", None, {}),
    ("hexagon_room", "hexagon_room.json", "0\nn\n", None, {"MCRT_DROPIN_CONTEXTS": "2"}),
    ("metals", "metals.json", "0\n", None, {"MCRT_DROPIN_CONTEXTS": "3"}),
    ("hexagon_room_pm", "hexagon_room.json", "0\ny\n", 4000, {}),
    ("hexagon_room_pm", "hexagon_room.json", "0\ny\n", 4000, {"MCRT_DROPIN_CONTEXTS": "2"}),
    ("hexagon_room_pm", "hexagon_room.json", "0\ny\n", 4000, {"MCRT_DROPIN_CONTEXTS": "3"}),
    ("hexagon_room_pm", "hexagon_room.json", "0\ny\n", 4000, {"MCRT_DROPIN_CONTEXTS": "2", "MCRT_DROPIN_REPLICATED_PHOTONS": "1"}),
    ("hexagon_room_pm", "hexagon_room.json", "0\ny\n", 4000, {"MCRT_DROPIN_CPU_PHOTONS": "1"})],
    ids=["hexagon_room", "hexagon_room-2ctx", "metals-3ctx", "hexagon_room_pm-gpu_photons", "hexagon_room_pm-gpu_photons-2ctx", "hexagon_room_pm-sharded_photons-3ctx",
         "hexagon_room_pm-replicated_photons-2ctx", "hexagon_room_pm-cpu_photons"])
def test_reference_main_renders_through_the_gpu(manifest, tmp_path, name, scene, answers, emissions, extra):
    if not os.path.exists(BIN) or not os.path.exists(os.path.join(SCENES, scene)):
        pytest.skip("oracle/_ref/mcrt_ref_gpu not built (python __graft_entry__.py build in the build container)")
    case = manifest["cases"][name]
    r = [x for x in case["renders"] if x.get("saves")][0]
    save = [s for s in r["saves"] if s["file"].endswith(".scene.tga")][0]  # Image::save with the scene file's own "image" object
    d, savename = _scene_dir(tmp_path, scene, r, emissions)
    dump = str(tmp_path / "frame.f64")
    env = dict(os.environ, MCRT_REF_SEED=str(manifest["seed"]), MCRT_DROPIN_DUMP=dump)
    env.update(extra)
    p = subprocess.run([BIN, "scenes"], input=answers, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "[mcrt_hip] Camera::sampleImage on the GPU" in p.stdout and "Render Completed" in p.stdout
    paths = r["width"] * r["height"] * r["sqrtspp"] ** 2
    assert "%d paths" % paths in p.stdout
    assert "%s context(s) on" % extra.get("MCRT_DROPIN_CONTEXTS", "1") in p.stdout
    if emissions:  # where the photon maps came from: the GPU pass unless the switch keeps the reference's constructor
        assert ("PhotonMapper pass on the GPU(s): %d emission paths" % (emissions * 10) in p.stdout) == ("MCRT_DROPIN_CPU_PHOTONS" not in extra)
        assert ("PHOTON MAPPING PASS" in p.stdout) == ("MCRT_DROPIN_CPU_PHOTONS" in extra)

    # the frame the GPU handed to camera.image vs the reference's own radiance
    frame = np.fromfile(dump).reshape(r["height"], r["width"], 3)
    ref = np.fromfile(golden_path(r["file"])).reshape(r["height"], r["width"], 3)
    rel = rel_error(frame, ref).max(axis=2)
    bad = int((rel > 1e-4).sum())
    print("%s: drop-in frame max rel %.3e, outliers %d / %d" % (name, rel.max(), bad, rel.size))
    assert bad <= max(2, int(0.002 * rel.size))

    # the file the reference's Image::save wrote from it vs the file it wrote from its own CPU render
    tga = np.fromfile(str(tmp_path / (savename + ".tga")), dtype=np.uint8)
    want = np.fromfile(golden_path(save["file"]), dtype=np.uint8)
    assert tga.shape == want.shape and np.array_equal(tga[:18], want[:18])  # HeaderTGA
    diff = np.abs(tga[18:].astype(int) - want[18:].astype(int))
    print("%s: %d of %d bytes differ (max step %d)" % (name, int((diff > 0).sum()), diff.size, int(diff.max())))
    # exposure and gain come from histograms of the whole frame, so last-ulp radiance differences (ocml vs glibc sin/cos) can move
    # a byte by one step where a value sits on an integer boundary; a wrong frame moves most bytes by many steps
    assert diff.max() <= 2 and (diff > 0).mean() < 0.02
