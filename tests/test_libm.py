"""csrc/mcrt_libm.hpp — glibc 2.35's sincos / sin / cos restated for the device — against the libm of the machine the test runs
on: bit equality on millions of arguments. This is what lets path-traced frames be compared with the reference bit for bit
(tests/test_gpu_parity.py): the reference's sin / cos pairs are sincos calls (the reference binary imports sincos, sincosf and
sin only), and sincos is glibc's baseline, FMA-free build of s_sincos.c, while sin alone (the Lanczos film filter) resolves to the
FMA build of s_sin.c on every x86-64 CPU with FMA + AVX2.

If the host's glibc is not 2.35-compatible in these routines, or its CPU lacks FMA (another IFUNC variant of sin), the comparison
is skipped with the reason - it cannot say anything about the restatement then."""
import ctypes as C
import math
import platform

import numpy as np
import pytest


def _has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


@pytest.mark.parametrize("n,lo,hi,edges", [(2000000, 0.0, 2.0 * math.pi, 1), (1000000, -7.0, 7.0, 0), (1000000, -1000.0, 1000.0, 0),
                                           (500000, -1.05e8, 1.05e8, 0), (500000, -1e-3, 1e-3, 0)])
def test_restated_sincos_sin_cos_equal_glibc(emu, n, lo, hi, edges):
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    emu.emu_libm_check.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_void_p]
    out = np.zeros(8, dtype=np.uint64)
    emu.emu_libm_check(n, 20260926, lo, hi, edges, out.ctypes.data)
    first = out[4:].view(np.float64)
    # sincos: the baseline build, the same on every x86-64 machine
    assert out[0] == 0 and out[1] == 0, "sincos differs, first at %r / %r" % (first[0], first[1])
    # sin / cos alone: the FMA variants
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: libm's sin / cos are another IFUNC variant than the one restated (sincos was checked)")
    assert out[2] == 0 and out[3] == 0, "sin / cos differ, first at %r / %r" % (first[2], first[3])


@pytest.mark.parametrize("n,lo,hi,edges", [(3000000, -1.0, 1.0, 1), (500000, 0.96, 1.0, 0), (500000, -0.13, 0.13, 0), (200000, -1e-7, 1e-7, 0),
                                           (100000, -1.5, 1.5, 0)])
def test_restated_asin_equals_glibc(emu, n, lo, hi, edges):
    """refAsin (Scene::skyColor's one libm call, scene.cpp:219-223) = the FMA variant of glibc 2.35's __ieee754_asin, bit for bit:
    dense over [-1, 1], the |x| >= 0.96875 branch, the polynomial branch, the tiny branch, every interval boundary, |x| > 1 (NaN)."""
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: libm's asin is another IFUNC variant than the one restated")
    emu.emu_asin_check.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_void_p]
    out = np.zeros(4, dtype=np.uint64)
    emu.emu_asin_check(n, 20260926, lo, hi, edges, out.ctypes.data)
    assert out[0] == 0, "asin differs on %d arguments, first at %r" % (out[0], out[1:2].view(np.float64)[0])


@pytest.mark.parametrize("n,scale", [(1500000, 0), (500000, 40)])
def test_restated_atan2_equals_glibc(emu, n, scale):
    """refAtan2 (the Photon constructor's two calls, photon.hpp:10-11) = the FMA variant of glibc 2.35's __ieee754_atan2, bit for bit:
    directions in every octant, unit-vector components as the emission pass passes them, ratios at every table-interval boundary,
    extreme ratios, zeros and axes, huge and subnormal magnitudes."""
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: libm's atan2 is another IFUNC variant than the one restated")
    emu.emu_atan2_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    out = np.zeros(3, dtype=np.uint64)
    emu.emu_atan2_check(n, 20260926, scale, out.ctypes.data)
    f = out[1:].view(np.float64)
    assert out[0] == 0, "atan2 differs on %d pairs, first at y = %r, x = %r" % (out[0], f[0], f[1])


def test_restated_sincosf_equals_glibc_on_every_float_of_the_range(emu):
    """refSinCosF (Photon::dir's sine / cosine pairs of two float angles, photon.hpp:19-27) = glibc 2.35's __sincosf_fma - and its sinf
    and cosf - bit for bit on EVERY float with |y| <= 4 (the photon angles are atan2 results, |y| <= pi: 2.1e9 arguments, all of them),
    and on every 4096th float from there to 120 (the end of the restated range)."""
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: libm's sincosf is another IFUNC variant than the one restated")
    emu.emu_sincosf_check.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.zeros(5, dtype=np.uint64)
    four = int(np.float32(4.0).view(np.uint32))
    emu.emu_sincosf_check(0, four, out.ctypes.data)
    assert not out[:4].any(), "sincosf / sinf / cosf differ on %s arguments of [-4, 4], first at bits %#x" % (out[:4], int(out[4]))
    top = int(np.float32(120.0).view(np.uint32)) - 1
    bad = np.zeros(4, dtype=np.uint64)
    for b in range(four, top, 4096):
        emu.emu_sincosf_check(b, b, out.ctypes.data)
        bad += out[:4]
    emu.emu_sincosf_check(top, top, out.ctypes.data)
    assert not (bad + out[:4]).any(), "sincosf differs on %s sampled arguments of (4, 120)" % (bad + out[:4])


def test_restated_pow_equals_glibc(emu):
    """refPow (sRGB::gammaCompress's std::pow(x, 1 / 2.4), color/srgb.hpp:54-62 - the one libm call of Image::save) = the FMA variant of
    glibc 2.35's pow, bit for bit: the gamma curve's own range densely, 600 binades of x with random exponents y, and the neighbourhoods of
    1 and of every boundary of the log table."""
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: libm's pow is another IFUNC variant than the one restated")
    emu.emu_pow_check.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_void_p]
    out = np.zeros(3, dtype=np.uint64)
    for family, n, lo, hi in ((0, 4000000, 0.0031308, 1.0), (0, 2000000, 1.0, 64.0), (0, 500000, 0.0, 0.0032), (1, 3000000, 0, 0), (2, 2000000, 0, 0)):
        emu.emu_pow_check(family, n, lo, hi, 20260927 + family, out.ctypes.data)
        f = out[1:].view(np.float64)
        assert out[0] == 0, "pow differs on %d of %d arguments of family %d, first at x = %r (%s), y = %r" % (out[0], n, family, f[0], float(f[0]).hex(), f[1])


@pytest.mark.gpu
def test_device_libm_equals_glibc_bits(pkg, emu):
    """The DEVICE's sincos / sin / cos / asin / atan2 (mcrt_libm through the C ABI: the functions the kernels inline, compiled for
    gfx950) against this host's glibc on 1.2e7 arguments: bit equality. This is the direct form of what the frame tests infer."""
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the restatement is of x86-64 glibc")
    if not (_has("fma") and _has("avx2")):
        pytest.skip("this CPU has no FMA/AVX2: its libm's sin / cos / asin / atan2 are other IFUNC variants than the ones restated")
    emu.emu_libm_host.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20260926)
    ctx = pkg.Context(0)

    def host(fn, a, b=None):
        o0, o1 = np.zeros_like(a), np.zeros_like(a)
        emu.emu_libm_host(fn, a.size, a.ctypes.data, b.ctypes.data if b is not None else None, o0.ctypes.data, o1.ctypes.data)
        return o0, o1

    def same(x, y):
        return np.array_equal(x.view(np.uint64), y.view(np.uint64))

    # sincos: the range the path uses (2 pi u), wider ranges, tiny arguments
    a = np.concatenate([rng.random(3000000) * 2.0 * math.pi, (rng.random(1000000) - 0.5) * 200.0, (rng.random(500000) - 0.5) * 2e8,
                        (rng.random(500000) - 0.5) * 1e-3])
    gs, gc = ctx.libm(pkg.LIBM_SINCOS, a)
    hs, hc = host(0, a)
    assert same(gs, hs) and same(gc, hc), "sincos: %d / %d arguments differ" % ((gs != hs).sum(), (gc != hc).sum())
    a = np.concatenate([rng.random(1000000) * 2.0 * math.pi, (rng.random(500000) - 0.5) * 2000.0])
    assert same(ctx.libm(pkg.LIBM_SIN, a), host(1, a)[0]), "sin"
    assert same(ctx.libm(pkg.LIBM_COS, a), host(2, a)[0]), "cos"
    # asin: dense over [-1, 1], the branch near 1, around every interval boundary
    edges = (32 + rng.integers(0, 225, 500000)) / 256.0 * (1.0 + (rng.integers(-4096, 4097, 500000)) * 2.0 ** -52)
    a = np.concatenate([rng.random(2000000) * 2.0 - 1.0, 1.0 - rng.random(500000) * 0.04, edges * np.where(rng.random(500000) < 0.5, -1.0, 1.0)])
    g, h = ctx.libm(pkg.LIBM_ASIN, a), host(3, a)[0]
    ok = (g.view(np.uint64) == h.view(np.uint64)) | (np.isnan(g) & np.isnan(h))
    assert ok.all(), "asin: %d arguments differ, first %r" % ((~ok).sum(), a[~ok][0])
    # atan2: directions in every octant; unit-vector components as Photon's constructor passes them; table-interval boundaries
    ang, r = (rng.random(1500000) * 2.0 - 1.0) * math.pi, np.exp2(rng.integers(-20, 21, 1500000)) * (1.0 + rng.random(1500000))
    d = rng.normal(size=(500000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    u = (1 + rng.integers(0, 512, 500000)) / 512.0 * (1.0 + rng.integers(-4096, 4097, 500000) * 2.0 ** -52)
    big = np.exp2(rng.integers(-10, 11, 500000)) * (1.0 + rng.random(500000))
    sw = rng.random(500000) < 0.5
    y = np.concatenate([r * np.sin(ang), np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2), d[:, 1], np.where(sw, big * u, big) * np.where(rng.random(500000) < 0.5, -1, 1)])
    x = np.concatenate([r * np.cos(ang), d[:, 2], d[:, 0], np.where(sw, big, big * u) * np.where(rng.random(500000) < 0.5, -1, 1)])
    g, h = ctx.libm(pkg.LIBM_ATAN2, y, x), host(4, y, x)[0]
    bad = g.view(np.uint64) != h.view(np.uint64)
    assert not bad.any(), "atan2: %d pairs differ, first y = %r, x = %r" % (bad.sum(), y[bad][0], x[bad][0])
    # sincosf (Photon::dir): float angles as the photon records hold them - atan2 results in [-pi, pi] - and wider / tiny arguments
    a = np.concatenate([(rng.random(3000000) * 2.0 - 1.0) * math.pi, (rng.random(500000) - 0.5) * 230.0, (rng.random(500000) - 0.5) * 1e-3,
                        (rng.random(200000) - 0.5) * 2.0 ** -11]).astype(np.float32).astype(np.float64)
    gs, gc = ctx.libm(pkg.LIBM_SINCOSF, a)
    hs, hc = host(5, a)
    assert same(gs, hs) and same(gc, hc), "sincosf: %d / %d arguments differ" % ((gs != hs).sum(), (gc != hc).sum())
    # pow (sRGB::gammaCompress: x^(1 / 2.4) over the tone-mapped range), other exponents over 600 binades (|y log x| < 512: the restated
    # main path - beyond it refPow hands over to the platform's pow), around 1 and the log table's boundaries
    g24 = 1.0 / 2.4
    near = (0x3fe6955500000000 + (rng.integers(0, 128, 500000).astype(np.uint64) << np.uint64(45)) + rng.integers(-4096, 4097, 500000).astype(np.uint64)).view(np.float64)
    x = np.concatenate([0.0031308 + rng.random(3000000) * 1.2, rng.random(500000) * 64.0, np.ldexp(1.0 + rng.random(1000000), rng.integers(-300, 301, 1000000)),
                        near, 1.0 + rng.integers(-4096, 4097, 200000) * 2.0 ** -52])
    y = np.concatenate([np.full(3500000, g24), rng.random(1000000) * 4.8 - 2.4, np.where(rng.random(500000) < 0.5, g24, rng.random(500000) * 4.8 - 2.4),
                        np.full(200000, g24)])
    g, h = ctx.libm(pkg.LIBM_POW, x, y), host(6, x, y)[0]
    bad = g.view(np.uint64) != h.view(np.uint64)
    assert not bad.any(), "pow: %d pairs differ, first x = %r (%s), y = %r" % (bad.sum(), x[bad][0], float(x[bad][0]).hex(), y[bad][0])
    ctx.close()


def test_pow_tables_are_glibcs():
    """The committed tables of refPow against an independent evaluation (Fractions, no libm): every log entry's logc + logctail is
    -log(invc) to 2^-68 with invc = j / 256 near its interval, ln 2 splits as the algorithm needs, every exp entry is 2^(i/128) to
    half an ulp with a tail below an ulp (tools/make_glibc_pow_tables.py, which found them in libm.so.6 by content)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_glibc_pow_tables", os.path.join(root, "tools", "make_glibc_pow_tables.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    logd, expd, etab = tool.committed()
    assert len(tool.verify(logd, expd, etab)) == 128


def test_sincos_table_is_glibcs():
    """The committed table against an independent evaluation of sin / cos at k/128 (high words must be the correctly rounded
    values; glibc's low words are within 2^-40 of the exact remainders, 17 of them not the nearest double)."""
    import os
    import struct
    from fractions import Fraction
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    words = []
    for line in open(os.path.join(root, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_glibc_sincostab.inc")):
        if line.strip().startswith("0x"):
            words += [int(t.strip().rstrip("ul"), 16) for t in line.strip().rstrip(",").split(",")]
    assert len(words) == 440
    tab = [struct.unpack("<d", struct.pack("<Q", w))[0] for w in words]

    def sin_cos(x, terms=30):
        s = c = Fraction(0)
        t = Fraction(1)
        for i in range(2 * terms):
            if i % 2 == 0:
                c += t if (i // 2) % 2 == 0 else -t
            else:
                s += t if (i // 2) % 2 == 0 else -t
            t = t * x / (i + 1)
        return s, c

    for k in range(110):
        s, c = sin_cos(Fraction(k, 128))
        for v, hi, lo in ((s, tab[4 * k], tab[4 * k + 1]), (c, tab[4 * k + 2], tab[4 * k + 3])):
            assert hi == float(v)
            rest = v - Fraction(hi)
            assert abs(Fraction(lo) - rest) <= abs(rest) * Fraction(1, 2 ** 40) + Fraction(1, 2 ** 1000)
