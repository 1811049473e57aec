"""Work-unit planning of the integrator launches (csrc/mcrt_plan.hpp, host build): a frame is cut into passes of rows that
fit the per-sample store, a pixel's samples into a power-of-two number of chunks. Every (row, sample) must be covered exactly
once whatever the sizes; the GPU tests (test_passes_and_chunks_do_not_change_the_frame) check the kernels that consume it."""
import ctypes as C
import itertools

import numpy as np
import pytest


def _plan(emu, width, rows, spp, store_gb, want):
    out = (C.c_uint64 * 4)()
    emu.emu_plan(width, rows, spp, store_gb, want, out)
    return dict(pass_rows=out[0], store_bytes=out[1], shift=out[2], chunk=out[3])


@pytest.mark.parametrize("width,rows,spp", [(1920, 1080, 256), (1920, 135, 256), (3840, 2160, 1024), (96, 54, 9), (40, 37, 1), (7, 3, 4),
                                            (1000, 1000, 256)])
def test_passes_cover_every_row_once_and_fit_the_store(emu, width, rows, spp):
    for store_gb in (16.0, 1.0, 0.0001):
        p = _plan(emu, width, rows, spp, store_gb, 1)
        pr = p["pass_rows"]
        assert pr >= 8 and pr % 8 == 0
        covered = np.zeros(rows, dtype=int)
        for base in range(0, rows, pr):
            end = min(rows, base + pr)
            covered[base:end] += 1
            assert (end - base) * width * spp * 24 <= p["store_bytes"]
        assert (covered == 1).all()
        # the store respects the budget unless 8 rows (the smallest pass) already exceed it
        assert p["store_bytes"] <= store_gb * 1e9 or pr == 8


def test_full_hd_frame_is_one_pass_and_4k_is_thirteen(emu):
    assert _plan(emu, 1920, 1080, 256, 16.0, 1)["pass_rows"] >= 1080              # 12.7 GB
    p = _plan(emu, 3840, 2160, 1024, 16.0, 1)                                       # 204 GB
    assert p["pass_rows"] == 168 and -(-2160 // p["pass_rows"]) == 13


@pytest.mark.parametrize("spp,want", list(itertools.product([1, 4, 9, 16, 64, 256, 1024, 1000], [0, 1, 2, 3, 16, 17, 100, 10 ** 6])))
def test_chunks_cover_every_sample_once(emu, spp, want):
    p = _plan(emu, 8, 8, spp, 16.0, want)
    units, chunk = 1 << p["shift"], p["chunk"]
    covered = np.zeros(spp, dtype=int)
    for c in range(units):                      # decodeUnit / wfShadeSlot: first = c * chunk, end = min(first + chunk, spp), skipped when empty
        first = c * chunk
        if first < spp:
            covered[first:min(first + chunk, spp)] += 1
    assert (covered == 1).all()
    assert chunk * units >= spp
    # as many units as asked for (rounded up to a power of two), unless chunks would drop below 4 samples
    assert units >= min(max(want, 1), max(1, spp // 4)) or (spp >> (p["shift"] + 1)) < 4
    assert p["shift"] == 0 or chunk >= 4 or spp < 8


@pytest.mark.parametrize("spp,lanes,pixels", list(itertools.product([1, 4, 16, 25, 64, 256, 1024], [131072, 262144], [1920 * 1080, 1920 * 136, 1000 * 1000, 192 * 108, 64])))
def test_megakernel_units_keep_sixteen_samples_unless_lanes_would_starve(emu, spp, lanes, pixels):
    """planChunksMega: 128 units per lane in chunks of >= 16 samples (a unit's fixed cost is about three path samples: 1/8 shards of
    the 1080p @ 256 spp frame 66.5 -> 57.2 ms); chunks down to 4 samples only when a lane would otherwise get fewer than 4 units."""
    out = (C.c_uint64 * 2)()
    emu.emu_plan_mega(spp, lanes, pixels, out)
    shift, chunk = int(out[0]), int(out[1])
    units = 1 << shift
    assert chunk * units >= spp and (shift == 0 or chunk * (units - 1) < spp + chunk)
    covered = np.zeros(spp, dtype=int)
    for c in range(units):
        first = c * chunk
        if first < spp:
            covered[first:min(first + chunk, spp)] += 1
    assert (covered == 1).all()
    per_lane = pixels * units / lanes
    if chunk < 16 and shift > 0:
        assert chunk >= 4 and pixels * (units // 2) < 4 * lanes   # short chunks only where chunks of 16+ starve the lanes
    if per_lane > 256 and shift > 0:
        assert pixels * (units // 2) < 128 * lanes               # never more units than the balance target asks for


def test_pipeline_pool_follows_the_pass(emu):
    """planPoolSlots: paths / 48 slots, at least 2.5 M, at most the cap (16 M by default), never fewer than 4 samples per slot, whole
    workgroups of the shade kernel - the sizes the A/B runs of round 4 found (profiles/r04_ab_pipeline_slot_paths.log)."""
    cap, block = 1 << 24, 256
    f = lambda paths, c=cap: int(emu.emu_plan_pool_slots(int(paths), c, block))
    assert f(1920 * 1080 * 1024) == cap                       # C3 at full size: the cap
    assert f(1920 * 1080 * 64) == 2764800                     # 133 M path samples: paths / 48
    assert f(1920 * 1080 * 16) == 2500096                     # 33 M: the floor (rounded up to whole workgroups)
    assert f(1920 * 1080 * 4) == 1920 * 1080                  # 8 M: four samples per slot
    assert f(1920 * 1080) == 1920 * 1080 // 4 + 0             # 2 M: four samples per slot
    assert f(1) == block and f(0) == block                    # never less than one workgroup
    assert f(10 ** 12, 1 << 20) == 1 << 20                    # a smaller cap (memory) wins
    for paths in (1, 255, 256, 1000, 10 ** 5, 10 ** 7, 10 ** 9, 10 ** 11):
        s = f(paths)
        assert s % block == 0 and block <= s <= cap
        assert s <= max(block, -(-max(paths // 4, 1) // block) * block)   # >= 4 samples per slot (up to the rounding to workgroups)
