#!/usr/bin/env python3
"""Writes monte-carlo-ray-tracer_amd/csrc/mcrt_glibc_sincostab.inc: the 440-entry table of glibc 2.35's double sin / cos
(sysdeps/ieee754/dbl-64/sincostab.c, __sincostab: for k = 0..109 the double-double values {sin(k/128) hi, lo, cos(k/128) hi, lo}).

glibc is a dependency of the REFERENCE (std::sin / std::cos in sampling.hpp:29-44, ggx.cpp:77-79, sphere.cpp:43) that is not in
/root/reference; the table is data of its published algorithm (IBM Accurate Mathematical Library, LGPL). It is read out of this
machine's libm (found by its leading entries, not by an address), and checked against an independent evaluation: every high
word must be the correctly rounded sin / cos of k/128 and every low word must agree with the exact remainder to 2^-40 relative
(17 of glibc's low words are not the nearest double - which is why the table cannot be regenerated from first principles).
"""
import os
import struct
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_glibc_sincostab.inc")


def sin_cos(x, terms=40):
    s = c = Fraction(0)
    t = Fraction(1)
    for n in range(2 * terms):
        if n % 2 == 0:
            c += t if (n // 2) % 2 == 0 else -t
        else:
            s += t if (n // 2) % 2 == 0 else -t
        t = t * x / (n + 1)
    return s, c


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
    blob = open(path, "rb").read()
    s1, c1 = sin_cos(Fraction(1, 128))
    lead = struct.pack("<4d", 0.0, 0.0, 1.0, 0.0) + struct.pack("<d", float(s1))
    at = blob.find(lead)
    if at < 0 or blob.find(lead, at + 1) >= 0:
        raise SystemExit("table not found (or not unique) in %s" % path)
    tab = struct.unpack_from("<440d", blob, at)
    for k in range(110):
        s, c = sin_cos(Fraction(k, 128))
        for v, hi, lo in ((s, tab[4 * k], tab[4 * k + 1]), (c, tab[4 * k + 2], tab[4 * k + 3])):
            assert hi == float(v), (k, hi)
            rest = v - Fraction(hi)
            assert abs(Fraction(lo) - rest) <= abs(rest) * Fraction(1, 2 ** 40) + Fraction(1, 2 ** 1000), (k, lo)
    with open(OUT, "w") as f:
        f.write("// glibc 2.35 __sincostab (sysdeps/ieee754/dbl-64/sincostab.c): {sin hi, sin lo, cos hi, cos lo} of k/128, k = 0..109,\n"
                "// as IEEE-754 bit patterns. Written by tools/make_glibc_sincos_table.py (which also checks them); do not edit.\n")
        for k in range(110):
            f.write("    " + ", ".join("0x%016xull" % struct.unpack("<Q", struct.pack("<d", tab[4 * k + j]))[0] for j in range(4)) + ",\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
