#!/usr/bin/env python3
"""Writes monte-carlo-ray-tracer_amd/csrc/mcrt_glibc_asintab.inc and mcrt_glibc_atantab.inc: the tables of glibc 2.35's double asin and atan2
(atan2: sysdeps/ieee754/dbl-64/e_atan2.c, table cij of uatan.tbl - the reference's Photon constructor, photon.hpp:10-11). asin:
(sysdeps/ieee754/dbl-64/e_asin.c, IBM Accurate Mathematical Library, LGPL-2.1-or-later): `asncs` (asincos.tbl, 2568 doubles: per
interval of |x| in [0.125, 0.96875) its centre x0, the Taylor coefficients of asin around x0 and asin(x0) as a double-double) and
`inroot` (root.tbl, 128 doubles: 1/sqrt seeds of the |x| >= 0.96875 branch). `powtwo` is 2^0 .. 2^26 and is computed, not stored.

glibc is a dependency of the REFERENCE (std::asin in Scene::skyColor, scene/scene.cpp:219-223) that is not in /root/reference; the
tables are data of its published algorithm. They are located in this machine's libm by CONTENT (the first interval's record, which
is a function of the algorithm: x0 = 0.126953125, f'(x0) = 1/sqrt(1 - x0^2), ...), not by an address, and checked independently:
for every interval the record's centre must lie in the interval, its first coefficient must be 1/sqrt(1 - x0^2) and its last two
words must add up to asin(x0), each to within an ulp of this Python's libm-free evaluation (Fractions + a series).
"""
import os
import struct
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_glibc_asintab.inc")
OUT_ATAN = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_glibc_atantab.inc")


def asin_frac(x, terms=400):
    """asin of a Fraction |x| < 0.97 as a Fraction: through atan(x / sqrt(1 - x^2)) would need a root; use the series of
    asin(x) = sum (2n)! / (4^n (n!)^2 (2n+1)) x^(2n+1), fine up to ~0.75; beyond that asin(x) = pi/2 - 2 asin(sqrt((1-x)/2))."""
    s = Fraction(0)
    t = x
    n = 0
    x2 = x * x
    while n < terms:
        s += t / (2 * n + 1)
        t = t * x2 * (2 * n + 1) / (2 * n + 2)
        n += 1
        if abs(t) < Fraction(1, 10 ** 40):
            break
    return s


def records():
    """(first index, record length, lower bound of |x|, upper bound) of every interval, as e_asin.c indexes them."""
    out = []
    for i in range(32):   # 0.125 <= |x| < 0.25: n = 11 * ((k & 0xfffff) >> 15), 2^-8 wide
        out.append((11 * i, 11, 0.125 + i / 256.0, 0.125 + (i + 1) / 256.0))
    for i in range(64):   # 0.25 <= |x| < 0.5: n = 11 * ((k & 0xfffff) >> 14) + 352
        out.append((352 + 11 * i, 11, 0.25 + i / 256.0, 0.25 + (i + 1) / 256.0))
    for i in range(64):   # 0.5 <= |x| < 0.75: n = 1056 + ((k & 0xfe000) >> 11) * 3 = 1056 + 12 i, 2^-8 wide
        out.append((1056 + 12 * i, 12, 0.5 + i / 256.0, 0.5 + (i + 1) / 256.0))
    for i in range(64, 108):  # 0.75 <= |x| < 0.921875: n = 992 + 13 ((k >> 13) & 0x7f)
        out.append((992 + 13 * i, 13, 0.5 + i / 256.0, 0.5 + (i + 1) / 256.0))
    for i in range(108, 116):  # < 0.953125: n = 884 + 14 i
        out.append((884 + 14 * i, 14, 0.5 + i / 256.0, 0.5 + (i + 1) / 256.0))
    for i in range(116, 120):  # < 0.96875: n = 768 + 15 i
        out.append((768 + 15 * i, 15, 0.5 + i / 256.0, 0.5 + (i + 1) / 256.0))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
    blob = open(path, "rb").read()
    lead = struct.pack("<d", 0.126953125)
    at = -1
    pos = blob.find(lead)
    while pos >= 0:  # the record of the first interval starts with its centre, followed by 1 / sqrt(1 - x0^2) = 1.00815728...
        nxt = struct.unpack_from("<d", blob, pos + 8)[0]
        if abs(nxt - 1.0081572852980196) < 1e-15:
            # (e_asin.c is compiled once per IFUNC variant - sse2, fma, fma4 - each with its own copy of the static table)
            if at >= 0 and blob[pos:pos + 2568 * 8] != blob[at:at + 2568 * 8]:
                raise SystemExit("two different asncs tables in %s" % path)
            if at < 0:
                at = pos
        pos = blob.find(lead, pos + 8)
    if at < 0:
        raise SystemExit("asncs table not found in %s" % path)
    asncs = struct.unpack_from("<2568d", blob, at)
    # inroot: 128 doubles starting 1.408721450121 (root.tbl: 1 / sqrt of the interval midpoints of [0.5, 2))
    lead = struct.pack("<2d", 1.408721450121, 1.39792649065766)
    ir = blob.find(lead)
    if ir < 0:
        raise SystemExit("inroot table not found in %s" % path)
    nx = blob.find(lead, ir + 8)
    while nx >= 0:
        if blob[nx:nx + 1024] != blob[ir:ir + 1024]:
            raise SystemExit("two different inroot tables in %s" % path)
        nx = blob.find(lead, nx + 8)
    inroot = struct.unpack_from("<128d", blob, ir)
    # -- checks
    covered = set()
    for n, ln, lo, hi in records():
        x0 = asncs[n]
        assert lo <= x0 <= hi, (n, x0, lo, hi)
        d1 = asncs[n + 1]
        # first coefficient = 1 / sqrt(1 - x0^2): compare squares, exactly
        f = Fraction(d1) ** 2 * (1 - Fraction(x0) ** 2)
        assert abs(f - 1) < Fraction(1, 2 ** 50), (n, d1)
        if x0 < 0.75:  # asin(x0) as {low word, high word} in front of the record's last two words (which e_asin.c does not read)
            val = asin_frac(Fraction(x0))
            two = Fraction(asncs[n + ln - 3]) + Fraction(asncs[n + ln - 4])
            assert abs(two - val) < Fraction(1, 2 ** 68), (n, float(two - val))  # (the low words carry ~30 bits)
        for k in range(ln):
            covered.add(n + k)
    # The end-to-end check is tests/test_libm.py: refAsin (which reads these tables) against the host's asin on millions of
    # arguments in every interval. Here also: every index of [0, 2568) belongs to a record, every word is finite.
    assert covered == set(range(2568)), (len(covered), sorted(set(range(2568)) - covered)[:10])
    for v in asncs + inroot:
        assert v == v and abs(v) < 1e16, v  # (the tenth-order coefficients near |x| = 0.97 reach 1e12)
    for i in range(128):  # inroot[i] ~ 1 / sqrt(m) for m the midpoint of the i-th 1/64-wide (i < 64: [0.5, 1) in 1/128 steps ...) interval
        assert 0.70 < inroot[i] < 1.42
    # atan2: cij (uatan.tbl), 241 rows of 7 doubles: x0 = the row's point, atan(x0), then the Taylor coefficients of atan around x0
    lead = struct.pack("<2d", 0.06347694384761945, 0.063391893014217)
    cj = blob.find(lead)
    if cj < 0:
        raise SystemExit("cij table not found in %s" % path)
    nx = blob.find(lead, cj + 8)
    while nx >= 0:
        if blob[nx:nx + 1687 * 8] != blob[cj:cj + 1687 * 8]:
            raise SystemExit("two different cij tables in %s" % path)
        nx = blob.find(lead, nx + 8)
    cij = struct.unpack_from("<1687d", blob, cj)
    import math
    for i in range(241):
        x0, a0, c2 = cij[7 * i], cij[7 * i + 1], cij[7 * i + 2]
        assert abs(x0 - (i + 16) / 256.0) < 1.0 / 256.0, (i, x0)           # the row for u with round(256 u) - 16 = i
        assert abs(a0 - math.atan(x0)) <= 2.3e-16 * a0, (i, a0)            # atan(x0) (this libm's, an ulp is enough: the KAT is test_libm)
        assert abs(c2 * (1.0 + x0 * x0) - 1.0) < 1e-15, (i, c2)            # first derivative 1 / (1 + x0^2)
    with open(OUT_ATAN, "w") as f:
        f.write("// glibc 2.35 e_atan2.c table cij (uatan.tbl, 241 rows x 7 words: x0, atan(x0), Taylor coefficients of atan around x0) as IEEE-754\n"
                "// bit patterns. IBM Accurate Mathematical Library, (C) Free Software Foundation, LGPL-2.1-or-later; read out of libm.so.6 by\n"
                "// tools/make_glibc_asin_atan_tables.py (which also checks the row structure); do not edit.\n")
        for i in range(0, 1687, 7):
            f.write("    " + ", ".join("0x%016xull" % struct.unpack("<Q", struct.pack("<d", w))[0] for w in cij[i:i + 7]) + ",\n")
    print("wrote", OUT_ATAN, "(cij at file offset 0x%x)" % cj)
    with open(OUT, "w") as f:
        f.write("// glibc 2.35 e_asin.c tables as IEEE-754 bit patterns: asncs (asincos.tbl, 2568 words) then inroot (root.tbl, 128 words).\n"
                "// IBM Accurate Mathematical Library, (C) Free Software Foundation, LGPL-2.1-or-later; read out of libm.so.6 by\n"
                "// tools/make_glibc_asin_atan_tables.py (which also checks the record structure); do not edit.\n")
        words = list(asncs) + list(inroot)
        for i in range(0, len(words), 4):
            f.write("    " + ", ".join("0x%016xull" % struct.unpack("<Q", struct.pack("<d", w))[0] for w in words[i:i + 4]) + ",\n")
    print("wrote", OUT, "(asncs at file offset 0x%x, inroot at 0x%x)" % (at, ir))


if __name__ == "__main__":
    main()
