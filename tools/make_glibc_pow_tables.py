#!/usr/bin/env python3
"""Writes monte-carlo-ray-tracer_amd/csrc/mcrt_glibc_powtab.inc: the data of glibc 2.35's double pow (sysdeps/ieee754/dbl-64/e_pow.c,
e_pow_log_data.c, e_exp_data.c - the "optimized routines" pow of Szabolcs Nagy / ARM, LGPL-2.1-or-later in glibc) - what
sRGB::gammaCompress (color/srgb.hpp:54-62, `std::pow(in[c], 1.0 / 2.4)`) calls for every byte Image::save writes.

  __pow_log_data   ln2hi, ln2lo, poly[7], tab[128] of {invc, pad, logc, logctail}
  __exp_data       invln2N, shift, negln2hiN, negln2loN, poly[4], (exp2 words, not used by pow), tab[2 * 128] of {tail bits, scale bits}

glibc is a dependency of the REFERENCE that is not in /root/reference; the tables are data of its published algorithm. They are located
in this machine's libm by CONTENT (the leading constants, which the algorithm fixes: ln2 split at 2^-45 steps followed by -0.5; 128 / ln2
followed by 0x1.8p52), not by an address, and checked independently of libm:
  * every log entry: invc is j / 128 or j / 256 spaced so that the c = 1 / invc cover [OFF, 2 OFF) = [0.7068, 1.4137), and
    logc + logctail = -log(invc) to 2^-70 (Fractions + an atanh series);
  * every exp entry: scale bits + (i << 45) read as a double is 2^(i/128) to half an ulp, and the tail is below an ulp of it.
The end-to-end check is tests/test_libm.py: refPow (which reads these tables) against the host's pow."""
import os
import struct
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "csrc", "mcrt_glibc_powtab.inc")


def log_frac(x, terms=200):
    """log of a Fraction x in [0.4, 2.5] as a Fraction: 2 atanh((x - 1) / (x + 1))."""
    t = (x - 1) / (x + 1)
    t2 = t * t
    s, p = Fraction(0), t
    for n in range(terms):
        s += p / (2 * n + 1)
        p *= t2
        if abs(p) < Fraction(1, 10 ** 45):
            break
    return 2 * s


def find_unique(blob, lead, length, what):
    at = blob.find(lead)
    if at < 0:
        raise SystemExit("%s not found" % what)
    nx = blob.find(lead, at + 8)
    while nx >= 0:  # (one copy per IFUNC variant of the function, if the data is static; these two are shared objects: one copy)
        if blob[nx:nx + length] != blob[at:at + length]:
            raise SystemExit("two different %s" % what)
        nx = blob.find(lead, nx + 8)
    return at


def verify(logd, expd, etab):
    """The structure checks, independent of libm: logd = ln2hi, ln2lo, poly[7], then 128 x {invc, pad, logc, logctail}; expd = invln2N,
    shift, negln2hiN, negln2loN, C2 .. C5; etab = 128 x {tail bits, scale bits}. Returns the 128 (invc, logc, logctail)."""
    assert logd[0] == float.fromhex("0x1.62e42fefa3800p-1") and logd[1] == float.fromhex("0x1.ef35793c76730p-45")
    poly = logd[2:9]
    # (e_pow_log_data.c: a degree-7 fit of log1p(r) - r, its coefficients scaled by -2 - A[0] * r * r is the -r^2 / 2 term)
    assert poly[0] == -0.5 and abs(poly[1] + 2.0 / 3.0) < 1e-12 and abs(poly[2] - 0.5) < 1e-12
    tab = []
    for i in range(128):
        invc, pad, logc, logctail = logd[9 + 4 * i: 13 + 4 * i]
        assert pad == 0.0
        c = 1.0 / invc
        # entry i serves the z whose bits lie in [OFF + (i << 45), OFF + ((i + 1) << 45)): c must be near them
        z_lo = struct.unpack("<d", struct.pack("<Q", 0x3fe6955500000000 + (i << 45)))[0]
        assert abs(c / z_lo - 1.0) < 1.0 / 64, (i, c, z_lo)
        f = Fraction(invc) * 256
        assert f.denominator == 1, (i, invc)          # invc = j / 256 (j / 128 for the lower half)
        val = -log_frac(Fraction(invc))
        assert abs(Fraction(logc) + Fraction(logctail) - val) < Fraction(1, 2 ** 68), (i, float(Fraction(logc) + Fraction(logctail) - val))
        tab.append((invc, logc, logctail))
    assert expd[0] == float.fromhex("0x1.71547652b82fep0") * 128 and expd[1] == float.fromhex("0x1.8p52")
    negln2hin, negln2lon = expd[2], expd[3]
    assert negln2hin == -float.fromhex("0x1.62e42fefa0000p-8") and abs(negln2lon + float.fromhex("0x1.cf79abc9e3b3ap-47")) < 1e-25
    # -(negln2hiN + negln2loN) * 128 = ln 2 to 2^-90
    assert abs(-(Fraction(negln2hin) + Fraction(negln2lon)) * 128 - log_frac(Fraction(2))) < Fraction(1, 2 ** 90)
    c2, c3, c4, c5 = expd[4:8]
    assert abs(c2 - 0.5) < 1e-12 and abs(c3 - 1 / 6.0) < 1e-12 and abs(c4 - 1 / 24.0) < 1e-7 and abs(c5 - 1 / 120.0) < 1e-7
    for i in range(128):
        tail, sbits = etab[2 * i], etab[2 * i + 1]
        scale = struct.unpack("<d", struct.pack("<Q", (sbits + (i << 45)) & 0xFFFFFFFFFFFFFFFF))[0]
        # 2^(i/128): compare its 128th power with 2^i, exactly
        r = Fraction(scale) ** 128 / Fraction(2) ** i
        assert abs(r - 1) < Fraction(128, 2 ** 52), (i, float(r - 1))
        t = struct.unpack("<d", struct.pack("<Q", tail))[0]
        assert abs(t) < 2.0 ** -52, (i, t)
    return tab


def committed():
    """(logd with the pad words put back, expd, etab) of the committed .inc."""
    import re
    text = open(OUT).read()
    arrays = {m.group(1): [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", m.group(2))]
              for m in re.finditer(r"MCRT_POWTAB_DECL\((\w+), \d+\) = \{(.*?)\};", text, re.S)}
    d = lambda w: struct.unpack("<d", struct.pack("<Q", w))[0]
    head, tab = arrays["kPowLogHead"], arrays["kPowLogTab"]
    assert len(head) == 9 and len(tab) == 384 and len(arrays["kPowExpHead"]) == 8 and len(arrays["kPowExpTab"]) == 256
    logd = [d(w) for w in head]
    for i in range(128):
        logd += [d(tab[3 * i]), 0.0, d(tab[3 * i + 1]), d(tab[3 * i + 2])]
    return logd, [d(w) for w in arrays["kPowExpHead"]], arrays["kPowExpTab"]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
    blob = open(path, "rb").read()
    # ---- __pow_log_data
    ln2hi, ln2lo = float.fromhex("0x1.62e42fefa3800p-1"), float.fromhex("0x1.ef35793c76730p-45")
    n_log = 9 + 128 * 4
    at = find_unique(blob, struct.pack("<3d", ln2hi, ln2lo, -0.5), n_log * 8, "__pow_log_data")
    logd = struct.unpack_from("<%dd" % n_log, blob, at)
    # ---- __exp_data
    invln2n, shift = float.fromhex("0x1.71547652b82fep0") * 128, float.fromhex("0x1.8p52")
    ea = find_unique(blob, struct.pack("<2d", invln2n, shift), 8 * 8, "__exp_data")
    expd = struct.unpack_from("<8d", blob, ea)
    # the table follows exp2shift and exp2_poly[5] (e_exp_data.c): located by ITS first entry {0, bits of 1.0}
    ta = blob.find(struct.pack("<2Q", 0, 0x3ff0000000000000), ea, ea + 0x100)
    if ta < 0:
        raise SystemExit("__exp_data.tab not found behind its constants")
    etab = struct.unpack_from("<256Q", blob, ta)
    tab = verify(logd, expd, etab)
    bits = lambda v: "0x%016xull" % struct.unpack("<Q", struct.pack("<d", v))[0]
    with open(OUT, "w") as f:
        f.write("// glibc 2.35 e_pow.c data as IEEE-754 bit patterns (e_pow_log_data.c, e_exp_data.c; (C) Free Software Foundation / ARM Ltd,\n"
                "// LGPL-2.1-or-later); read out of libm.so.6 by tools/make_glibc_pow_tables.py (which also checks the entries); do not edit.\n"
                "// kPowLogHead: ln2hi, ln2lo, poly[7].  kPowLogTab: 128 x {invc, logc, logctail}.  kPowExpHead: invln2N, shift, negln2hiN,\n"
                "// negln2loN, C2 .. C5.  kPowExpTab: 128 x {tail bits, scale bits}.\n")
        f.write("MCRT_POWTAB_DECL(kPowLogHead, 9) = {\n    " + ", ".join(bits(v) for v in logd[:9]) + "};\n")
        f.write("MCRT_POWTAB_DECL(kPowLogTab, 384) = {\n")
        for invc, logc, logctail in tab:
            f.write("    %s, %s, %s,\n" % (bits(invc), bits(logc), bits(logctail)))
        f.write("};\n")
        f.write("MCRT_POWTAB_DECL(kPowExpHead, 8) = {\n    " + ", ".join(bits(v) for v in expd) + "};\n")
        f.write("MCRT_POWTAB_DECL(kPowExpTab, 256) = {\n")
        for i in range(0, 256, 4):
            f.write("    " + ", ".join("0x%016xull" % w for w in etab[i:i + 4]) + ",\n")
        f.write("};\n")
    print("wrote", OUT, "(__pow_log_data at file offset 0x%x, __exp_data at 0x%x, its table at 0x%x)" % (at, ea, ta))


if __name__ == "__main__":
    main()
