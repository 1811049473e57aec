// pow of glibc 2.35, bit for bit — the one libm call of the output stage: sRGB::gammaCompress's std::pow(in[c], 1.0 / 2.4)
// (color/srgb.hpp:54-62), three calls per pixel of Image::save (camera/image.cpp:47). With the platform's pow (ocml: another
// algorithm, last-bit differences) a byte of the .tga moved wherever the product landed within an ulp of a rounding boundary:
// at most one LSB on fewer than 1e-4 of the bytes through round 4. With this function the bytes are the reference's.
//
// glibc's pow (sysdeps/ieee754/dbl-64/e_pow.c; Szabolcs Nagy's "optimized routines" pow) is log(x) as a double-double through a
// 128-entry table and a degree-7 polynomial (log_inline), times y as a double-double (ehi, elo), into exp through a 128-entry table
// of 2^(i/128) and a degree-5 polynomial (exp_inline). std::pow is an IFUNC (sysdeps/x86_64/fpu/multiarch/e_pow.c): on a CPU with
// FMA and AVX2 - the build container's and the GPU box's - the dynamic linker picks __pow_fma, the same source compiled with -mfma,
// where the source's own __builtin_fma calls AND every a * b + c the compiler contracted are one vfmadd. refPow restates THAT
// instruction sequence (Ubuntu glibc 2.35-0ubuntu3.11, read off its disassembly; each fmaD below is one vfmadd there, each other
// operation is rounded on its own). Only the main path is restated - x positive, finite and normal, y of ordinary magnitude,
// |y log x| < 512 - which is all gammaCompress can ask for; anything else (zeros, negatives, infinities, NaNs, subnormals,
// overflow, underflow) goes to the platform's pow. Tables: mcrt_glibc_powtab.inc (tools/make_glibc_pow_tables.py).
// tests/test_libm.py: bit-equal to the host's pow on millions of arguments (x over the gamma curve's range and over 600 binades, y =
// 1 / 2.4 and random exponents), on the host and on the GPU.
//
// ATTRIBUTION: as for mcrt_libm.hpp - this header and its table restate / are data of the GNU C Library 2.35 (e_pow.c, e_pow_log_data.c,
// e_exp_data.c: Copyright (C) Free Software Foundation, Inc. / ARM Ltd., LGPL-2.1-or-later) and are offered under the same terms;
// -DMCRT_PLATFORM_LIBM builds without them (refPow = the platform's pow; .tga bytes then within one LSB of the reference's).
#pragma once

#include "mcrt_libm.hpp"

namespace mcrt {

#if defined(MCRT_PLATFORM_LIBM)
MCRT_HD double refPow(double x, double y) { return pow(x, y); }
#else

namespace glibc235 {
#define MCRT_POWTAB_DECL(name, n) MCRT_LIBM_TABLE unsigned long long name[n]
#include "mcrt_glibc_powtab.inc"
#undef MCRT_POWTAB_DECL
}  // namespace glibc235

MCRT_HD double refPow(double x, double y) {
    using namespace glibc235;
    constexpr bool F = true;
    const unsigned long long ix = dBits(x), iy = dBits(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
    // e_pow.c:289-290: anything but "x positive normal finite, 2^-65 <= |y| < 2^63" is the special-case code there
    if (!(topx - 0x001u < 0x7ffu - 0x001u && (topy & 0x7ffu) - 0x3beu < 0x43eu - 0x3beu)) return pow(x, y);

    // ---- log_inline (e_pow.c:38-112): x = 2^k z, z in [OFF, 2 OFF), c near z from the table, r = z / c - 1 exactly
    const unsigned long long tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((long long)tmp >> 52);
    const double z = bitsD(ix - (tmp & 0xfff0000000000000ull));
    const double kd = (double)k;
    const double invc = bitsD(kPowLogTab[3 * i]), logc = bitsD(kPowLogTab[3 * i + 1]), logctail = bitsD(kPowLogTab[3 * i + 2]);
    const double Ln2hi = bitsD(kPowLogHead[0]), Ln2lo = bitsD(kPowLogHead[1]);
    const double A0 = bitsD(kPowLogHead[2]), A1 = bitsD(kPowLogHead[3]), A2 = bitsD(kPowLogHead[4]), A3 = bitsD(kPowLogHead[5]),
                 A4 = bitsD(kPowLogHead[6]), A5 = bitsD(kPowLogHead[7]), A6 = bitsD(kPowLogHead[8]);
    const double t1 = fmaD<F>(kd, Ln2hi, logc);
    const double r = fmaD<F>(z, invc, -1.0);
    const double ar = r * A0;
    const double lo1 = fmaD<F>(kd, Ln2lo, logctail);
    const double q12 = fmaD<F>(r, A2, A1);
    const double q34 = fmaD<F>(r, A4, A3);
    const double t2 = r + t1;
    const double ar2 = r * ar;
    const double lo2 = (t1 - t2) + r;
    const double ar3 = r * ar2;
    const double lo3 = fmaD<F>(ar, r, -ar2);
    const double q56 = fmaD<F>(r, A6, A5);
    const double hi = t2 + ar2;
    const double lo4 = (t2 - hi) + ar2;
    const double poly = fmaD<F>(ar2, fmaD<F>(q56, ar2, q34), q12);
    const double lo = fmaD<F>(ar3, poly, ((lo1 + lo2) + lo3) + lo4);
    const double lhi = hi + lo;
    const double llo = (hi - lhi) + lo;

    // ---- y log(x) as ehi + elo (e_pow.c:340-343)
    const double ehi = y * lhi;
    const double elo = fmaD<F>(y, llo, fmaD<F>(lhi, y, -ehi));

    // ---- exp_inline (e_pow.c:170-232), sign_bias 0
    const uint32_t abstop = (uint32_t)(dBits(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x03fu) {
        if (abstop < 0x3c9u) return 1.0 + ehi;  // |y log x| < 2^-54 (e_pow.c:186-189, WANT_ROUNDING)
        return pow(x, y);                        // |y log x| >= 512: its overflow / underflow code
    }
    const double InvLn2N = bitsD(kPowExpHead[0]), Shift = bitsD(kPowExpHead[1]), NegLn2hiN = bitsD(kPowExpHead[2]),
                 NegLn2loN = bitsD(kPowExpHead[3]);
    const double C2 = bitsD(kPowExpHead[4]), C3 = bitsD(kPowExpHead[5]), C4 = bitsD(kPowExpHead[6]), C5 = bitsD(kPowExpHead[7]);
    double kd2 = fmaD<F>(ehi, InvLn2N, Shift);
    const unsigned long long ki = dBits(kd2);
    kd2 = kd2 - Shift;
    double rr = fmaD<F>(kd2, NegLn2loN, fmaD<F>(kd2, NegLn2hiN, ehi));
    rr = elo + rr;
    const uint32_t idx = 2u * (uint32_t)(ki & 127u);
    const unsigned long long sbits = kPowExpTab[idx + 1] + (ki << 45);
    const double tail = bitsD(kPowExpTab[idx]);
    const double p23 = fmaD<F>(rr, C3, C2);
    const double base = rr + tail;
    const double r2 = rr * rr;
    const double p45 = fmaD<F>(rr, C5, C4);
    const double part = fmaD<F>(p23, r2, base);
    const double tmpv = fmaD<F>(p45, r2 * r2, part);
    const double scale = bitsD(sbits);
    return fmaD<F>(tmpv, scale, scale);
}
#endif  // MCRT_PLATFORM_LIBM

}  // namespace mcrt
