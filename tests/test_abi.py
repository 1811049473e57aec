"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/mcrt.h declares,
refuses to run without a device (no CPU fallback), and the host-only helpers behave."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, _has_gpu, golden_path


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcrt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libmcrt_hip.so does not export %s" % n


def test_struct_sizes_match_header(pkg, tmp_path):
    # compile include/mcrt.h with the C compiler and compare the layouts the binding assumes
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "mcrt.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(mcrt_material),sizeof(mcrt_scene_desc),sizeof(mcrt_photon_map_desc),sizeof(mcrt_camera_desc),'
                   'sizeof(mcrt_stats),offsetof(mcrt_scene_desc,scene_ior));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes[:5] == [C.sizeof(pkg.Material), C.sizeof(pkg.SceneDesc), C.sizeof(pkg.PhotonMapDesc),
                         C.sizeof(pkg.CameraDesc), C.sizeof(pkg.Stats)]
    assert sizes[5] == pkg.SceneDesc.scene_ior.offset


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device error path")
def test_create_fails_loudly_without_device(pkg):
    with pytest.raises(pkg.McrtError) as e:
        pkg.Context(0)
    assert "-2" in str(e.value) or "no HIP device" in str(e.value)


def test_image_roundtrip_and_params(pkg, manifest, tmp_path):
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    s = img.scene
    assert s.num_surfaces == 44 and s.num_nodes == 16 and s.num_lights == 2 and s.scene_ior == 1.75
    assert img.param("k_nearest_photons") == 50 and img.param("photon_mapping") == 1
    assert img.param("global_seed") == manifest["seed"]
    assert img.photons(0).num_photons > 1000 and img.photons(1).num_photons > 1000
    L = pkg.lib()
    out = str(tmp_path / "copy.mcrt")
    L.mcrt_image_save.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    cam = img.camera
    keys = (C.c_char_p * 2)(b"k_nearest_photons", b"global_seed")
    vals = (C.c_uint64 * 2)(50, manifest["seed"])
    rc = L.mcrt_image_save(out.encode(), C.byref(s), C.byref(cam), C.byref(img.photons(0)), C.byref(img.photons(1)), keys, vals, 2)
    assert rc == 0
    img2 = pkg.SceneImage(out)
    s2 = img2.scene
    assert (s2.num_nodes, s2.num_surfaces, s2.num_materials, s2.num_lights) == (s.num_nodes, s.num_surfaces, s.num_materials, s.num_lights)
    a = np.ctypeslib.as_array(s.surf_v, shape=(s.num_surfaces * 9,))
    b = np.ctypeslib.as_array(s2.surf_v, shape=(s.num_surfaces * 9,))
    np.testing.assert_array_equal(a, b)
    assert img2.photons(1).num_photons == img.photons(1).num_photons
    assert img2.camera.width == cam.width


def test_image_load_errors(pkg, tmp_path):
    with pytest.raises(pkg.McrtError):
        pkg.SceneImage(str(tmp_path / "missing.mcrt"))
    bad = tmp_path / "bad.mcrt"
    bad.write_bytes(b"not an image")
    with pytest.raises(pkg.McrtError):
        pkg.SceneImage(str(bad))


def test_shard_rows_partition(pkg):
    cam = pkg.CameraDesc()
    cam.width, cam.height, cam.sqrtspp = 100, 77, 1
    cam.shard_count = 1
    np.testing.assert_array_equal(pkg.shard_rows(cam), np.arange(77))
    for count, group in ((2, 8), (8, 32), (3, 5), (8, 1)):
        seen = []
        for i in range(count):
            cam.shard_index, cam.shard_count, cam.shard_rows = i, count, group
            rows = pkg.shard_rows(cam)
            assert np.all((rows // group) % count == i)
            seen.append(rows)
        allrows = np.sort(np.concatenate(seen))
        np.testing.assert_array_equal(allrows, np.arange(77))  # ragged last group included exactly once
