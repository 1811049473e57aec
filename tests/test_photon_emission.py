"""Photon emission pass: PhotonMapper's per-emission set-up and emitPhoton
(integrator/photon-mapper/photon-mapper.cpp:24-115, 225-277). The reference's result is the photon
content of the maps in tests/golden/hexagon_room_pm.mcrt (flattened from the reference's own
LinearOctree<Photon>::ordered_data; 4000 emissions x caustic_factor 10 = 40 000 photon paths)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_oracle_bits, golden_path, host_libm_is_the_restated_one, photon_set_bytes, sort_by_key

EMISSIONS, CAUSTIC_FACTOR = 4000, 10.0


def _reference_maps(img):
    out = []
    for which in (0, 1):
        pm = img.photons(which)
        n = pm.num_photons
        out.append(np.ctypeslib.as_array(pm.photons, shape=(n * 8,)).reshape(n, 8).copy())
    return out


def test_oracle_emission_equals_reference_photon_sets(pkg, oracle, manifest):
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    r = oracle.emit_photons(img, EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    ref_g, ref_c = _reference_maps(img)
    assert r["paths"] == 40000
    assert np.array_equal(photon_set_bytes(r["global_"][0]), photon_set_bytes(ref_g))
    assert np.array_equal(photon_set_bytes(r["caustic"][0]), photon_set_bytes(ref_c))
    # keys are unique and ordered (light, emission, bounce)
    for name in ("global_", "caustic"):
        k = r[name][1]
        assert np.all(np.diff(k.astype(np.int64)) > 0)


@pytest.mark.parametrize("stage_all", [0, 1])
def test_emission_device_code_equals_oracle(pkg, emu, oracle, manifest, stage_all):
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    want = oracle.emit_photons(img, EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    cap = 1 << 17
    g, gk = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
    c, ck = np.zeros((cap, 8), dtype=np.float32), np.zeros(cap, dtype=np.uint64)
    ng, nc, rays = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = emu.emu_emit_photons(C.byref(img.scene), float(EMISSIONS), CAUSTIC_FACTOR, manifest["seed"], stage_all, g.ctypes.data,
                              gk.ctypes.data, cap, C.byref(ng), c.ctypes.data, ck.ctypes.data, cap, C.byref(nc), C.byref(rays))
    assert rc == 0
    for (got, gotk, n), name in (((g, gk, ng.value), "global_"), ((c, ck, nc.value), "caustic")):
        a, ak = sort_by_key(got[:n], gotk[:n])
        b, bk = want[name]
        np.testing.assert_array_equal(ak, bk)
        assert_oracle_bits(a, b)  # same libm on the host: same bits
    assert rays.value == want["rays"]


@pytest.mark.gpu
def test_gpu_emission_matches_oracle(pkg, oracle, manifest):
    """Through the C ABI on the GPU: the photon LISTS are the oracle's - and so the reference's (test above) - bit for bit, matched
    by key (light, emission index, bounce). Every libm call of a photon path is restated for the device (csrc/mcrt_libm.hpp: sincos
    since round 3, atan2 of the stored direction since round 4), so there is no tolerance and no allowance for unmatched keys."""
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    want = oracle.emit_photons(img, EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    ctx = pkg.Context(0)
    ctx.upload_scene(img.scene)
    got = ctx.emit_photons(EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    assert got["paths"] == want["paths"] == 40000
    assert got["rays"] == want["rays"]
    for name in ("global_", "caustic"):
        a, ak = sort_by_key(*got[name])
        b, bk = want[name]
        np.testing.assert_array_equal(ak, bk, err_msg="%s: photon keys" % name)
        assert_oracle_bits(a, b, "%s: photon records" % name)  # (bits where the oracle runs on the restated libm, 2e-6 elsewhere)
    # ... and as sets they are the photon content of the reference's own maps (reference-made fixture: host-independent)
    ref_g, ref_c = _reference_maps(img)
    assert np.array_equal(photon_set_bytes(got["global_"][0]), photon_set_bytes(ref_g))
    assert np.array_equal(photon_set_bytes(got["caustic"][0]), photon_set_bytes(ref_c))
    ctx.close()


@pytest.mark.gpu
def test_gpu_emission_shards_union_equals_whole(pkg, manifest):
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    ctx = pkg.Context(0)
    ctx.upload_scene(img.scene)
    whole = ctx.emit_photons(EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    parts = [ctx.emit_photons(EMISSIONS, CAUSTIC_FACTOR, manifest["seed"], i, 3) for i in range(3)]
    assert sum(p["paths"] for p in parts) == whole["paths"]
    for name in ("global_", "caustic"):
        ph = np.concatenate([p[name][0] for p in parts])
        keys = np.concatenate([p[name][1] for p in parts])
        a, ak = sort_by_key(ph, keys)
        b, bk = sort_by_key(*whole[name])
        np.testing.assert_array_equal(ak, bk)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    ctx.close()


@pytest.mark.gpu
def test_gpu_emission_list_growth_and_errors(pkg, manifest):
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    ctx = pkg.Context(0)
    with pytest.raises(pkg.McrtError):
        ctx.emit_photons(100, 10.0, 1)  # no scene yet
    ctx.upload_scene(img.scene)
    small = ctx.emit_photons(100, 10.0, manifest["seed"])
    assert small["paths"] == 1000 and len(small["caustic"][0]) > 0
    with pytest.raises(pkg.McrtError):
        ctx.emit_photons(100, 0.0, 1)
    ctx.close()


def _as_np(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy() if n else np.zeros(0, dtype=dtype)


def test_host_map_builder_reproduces_reference_octree(pkg, oracle, manifest):
    """mcrt_photon_map_build on the reference's own photon lists gives the reference's LinearOctree:
    same octants (tight boxes, leaf flags, counts, links, data ranges); only the photon order inside a
    leaf may differ."""
    img = pkg.SceneImage(golden_path("hexagon_room_pm.mcrt"))
    s = img.scene
    for which in (0, 1):
        ref = img.photons(which)
        n, no = ref.num_photons, ref.num_octants
        photons = np.ctypeslib.as_array(ref.photons, shape=(n * 8,)).reshape(n, 8).copy()
        rng = np.random.default_rng(which)
        built = pkg.PhotonMap(photons[rng.permutation(n)], s.bb_min[:], s.bb_max[:], 200)
        d = built.desc
        assert d.num_octants == no and d.num_photons == n
        for field, dt, k in (("octant_bounds", np.float64, 6), ("octant_start_data", np.uint64, 1),
                             ("octant_contained_data", np.uint64, 1), ("octant_next_sibling", np.uint32, 1), ("octant_leaf", np.uint8, 1)):
            np.testing.assert_array_equal(_as_np(getattr(d, field), no * k, dt), _as_np(getattr(ref, field), no * k, dt), err_msg=field)
        got = np.ctypeslib.as_array(d.photons, shape=(n * 8,)).reshape(n, 8)
        start, cont, leaf = (_as_np(d.octant_start_data, no, np.int64), _as_np(d.octant_contained_data, no, np.int64),
                             _as_np(d.octant_leaf, no, np.uint8))
        for o in np.nonzero(leaf)[0]:  # every leaf holds the same photons
            a = photon_set_bytes(got[start[o]:start[o] + cont[o]])
            b = photon_set_bytes(photons[start[o]:start[o] + cont[o]])
            assert np.array_equal(a, b)
        # and queries agree with the reference map
        pts = np.ctypeslib.as_array(ref.photons, shape=(n * 8,)).reshape(n, 8)[:200, 3:6].astype(np.float64) + 0.01
        c0, i0, d0 = oracle.knn(ref, pts, 50)
        c1, i1, d1 = oracle.knn(d, pts, 50)
        np.testing.assert_array_equal(c0, c1)
        np.testing.assert_array_equal(d0, d1)


@pytest.mark.gpu
def test_gpu_photon_mapping_end_to_end(pkg, manifest):
    """emit on the GPU -> build the maps on the host -> upload -> photon-mapped render on the GPU, against
    the reference's image (rendered by the reference from its own emission pass and octrees)."""
    from conftest import camera_for, load_radiance, rel_error
    case = manifest["cases"]["hexagon_room_pm"]
    img = pkg.SceneImage(golden_path(case["image"]))
    s = img.scene
    ctx = pkg.Context(0)
    ctx.upload_scene(s)
    em = ctx.emit_photons(EMISSIONS, CAUSTIC_FACTOR, manifest["seed"])
    gmap = pkg.PhotonMap(em["global_"][0], s.bb_min[:], s.bb_max[:], 200)
    cmap = pkg.PhotonMap(em["caustic"][0], s.bb_min[:], s.bb_max[:], 200)
    ctx.upload_photons(gmap.desc, cmap.desc, 50, False)
    r = case["renders"][0]
    out, st = ctx.sample_image(camera_for(img, r), manifest["seed"], pkg.INTEGRATOR_PHOTON_MAPPER)
    rel = rel_error(out, load_radiance(r)).max(axis=2)
    bad = int((rel > 1e-4).sum())
    print("end-to-end photon mapping: max rel %.3e, outliers %d / %d, %d kNN searches" % (rel.max(), bad, rel.size, st["knn_searches"]))
    assert bad <= max(2, int(0.002 * rel.size))
    ctx.close()
