// sin and cos as the REFERENCE computes them - bit for bit.
//
// Where the reference takes the sine AND the cosine of one angle - every such call on the path: sampling/sampling.hpp:29-44
// (uniform disk, cosWeightedHemi), material/ggx.cpp:77-79, surface/sphere.cpp:43 - g++ merges the pair into ONE call of glibc's
// sincos (its cse_sincos pass; `nm -D` of the reference binary: it imports sincos, sincosf and sin, and no cos at all). glibc
// 2.35 (sysdeps/ieee754/dbl-64/s_sincos.c + s_sin.c, IBM Accurate Mathematical Library; a dependency of the reference that is
// not in /root/reference) builds sincos from the inline kernels do_sin / do_cos / TAYLOR_SIN / reduce_sincos of s_sin.c. ocml's
// sin / cos differ from it in the last bit for a few arguments in ten thousand, and one such bit can send one of a pixel's 256
// paths down another branch: the only source of per-pixel outliers in path-traced frames of rounds 1 and 2 (every +, -, x, /,
// sqrt is IEEE-exact on gfx950 and the code is built with -ffp-contract=off). This header restates glibc's algorithm so that
// the device produces glibc's bits.
//
// WHICH code, exactly. x86-64 glibc has two compilations of those kernels:
//   * sincos is an ordinary function built for baseline x86-64 (no FMA instruction in it): the C source as written, every
//     operation rounded - refSinCos below (kFused = false);
//   * sin and cos are IFUNCs (sysdeps/x86_64/fpu/multiarch/s_sin.c): on a CPU with FMA and AVX2 - the build container and the
//     GPU box's EPYC included - the dynamic linker selects __sin_fma / __cos_fma, the same source compiled with -mfma, in which
//     the compiler contracted a*b + c into fused multiply-adds. The reference calls sin alone once (the Lanczos film filter,
//     camera/filter.hpp:64): refSin (kFused = true) restates those contractions, read off the instruction sequence of Ubuntu's
//     glibc 2.35-0ubuntu3.11 (every fused a*b + c below is an fma there; with kFused = false the same expression is the
//     source's two roundings). The two results differ for about one argument in a thousand.
// tests/test_libm.py checks all three against the running host's libm (sincos, sin, cos called separately) on millions of
// arguments - dense in [0, 2 pi], the range the path uses (2 pi u, u in [0, 1)), wider ranges, the branch boundaries.
// Table: mcrt_glibc_sincostab.inc (tools/make_glibc_sincos_table.py).
//
// Range: |x| < 105414350 (the Cody-Waite reduction of reduce_sincos); beyond that - never reached by the path - the platform's
// sin / cos answer.
//
// ATTRIBUTION. refSinCos / refSin / refCos / refAsin / refAtan2 restate algorithms of the GNU C Library 2.35 (sysdeps/ieee754/dbl-64:
// s_sin.c, s_sincos.c, e_asin.c, e_atan2.c and their tables sincostab.c, asincos.tbl, root.tbl, uatan.tbl - the IBM Accurate Mathematical
// Library, Copyright (C) Free Software Foundation, Inc., licensed LGPL-2.1-or-later). The three .inc tables next to this file are data
// of that library, read out of the build machine's libm.so.6 by tools/make_glibc_*_table*.py; this header and those tables are offered
// under the same terms (LGPL-2.1-or-later). The rest of the repository does not depend on them for anything but bit parity with a
// reference that links glibc: replacing the five functions by the platform's sin / cos / asin / atan2 keeps every frame within 1e-12.
//
// BUILD SWITCH. -DMCRT_PLATFORM_LIBM (MCRT_PLATFORM_LIBM=1 python -m monte-carlo-ray-tracer_amd.build) compiles NONE of the above: the six
// functions call the platform's libm (ocml on the device), the tables are not included and the library carries no LGPL component. Frames
// then agree with the reference to 1e-12 instead of bit for bit (tests that demand the reference's bits fail by design in such a build).
#pragma once

#include "mcrt_math.hpp"

#if defined(MCRT_PLATFORM_LIBM)
namespace mcrt {
namespace glibc235 {
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const unsigned long long kSinCosTab[440] = {0ull};  // (the kernels' staging code is written for a table of this size)
__device__ inline void stageSinCosTab() {}
__device__ inline void ldsTabStore(uint32_t, unsigned long long) {}
#else
static const unsigned long long kSinCosTab[440] = {0ull};
inline void stageSinCosTab() {}
inline void ldsTabStore(uint32_t, unsigned long long) {}
#endif
constexpr uint32_t kShadeStaticLds = 0u;
}  // namespace glibc235
MCRT_HD double refSin(double x) { return sin(x); }
MCRT_HD double refCos(double x) { return cos(x); }
template <bool kLds = true>
MCRT_HD void refSinCos(double x, double& sn_out, double& cs_out) {
    sn_out = sin(x);
    cs_out = cos(x);
}
MCRT_HD double refAsin(double x) { return asin(x); }
MCRT_HD double refAtan2(double y, double x) { return atan2(y, x); }
MCRT_HD void refSinCosF(float y, float& sn_out, float& cs_out) {
    sn_out = sinf(y);
    cs_out = cosf(y);
}
}  // namespace mcrt
#else

namespace mcrt {
namespace glibc235 {

#if defined(__HIP_DEVICE_COMPILE__)
#define MCRT_LIBM_TABLE __device__ const
#else
#define MCRT_LIBM_TABLE static const
#endif
MCRT_LIBM_TABLE unsigned long long kSinCosTab[440] = {
#include "mcrt_glibc_sincostab.inc"
};

// On the GPU the table is read from LDS: a lookup in global memory is one more dependent round trip per call (two or three calls per
// bounce; the shading kernels are bound by exactly such chains). Every kernel that shades calls stageSinCosTab() before its first
// barrier; the 3520 bytes are static LDS of those kernels only (kShadeStaticLds, taken off their dynamic LDS budget by the host).
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned long long ldsSinCosTab[440];
__device__ inline void stageSinCosTab() {
    for (uint32_t i = threadIdx.x; i < 440u; i += blockDim.x) ldsSinCosTab[i] = kSinCosTab[i];
}
// kLds = false: from memory (the kernels whose whole scene is LDS-resident: their LDS pipe is the busy one; measured on the C2
// frame 446.5 ms against 448.5 with the table in LDS, the lane state machine 329 -> 317 ms the other way round)
__device__ inline void ldsTabStore(uint32_t i, unsigned long long v) { ldsSinCosTab[i] = v; }  // (for kernels that stage by hand)
template <bool kLds = true>
MCRT_HD double tabAt(int i) { return bitsD(kLds ? ldsSinCosTab[i] : kSinCosTab[i]); }
#else
inline void stageSinCosTab() {}
inline void ldsTabStore(uint32_t, unsigned long long) {}
template <bool kLds = true>
MCRT_HD double tabAt(int i) { return bitsD(kSinCosTab[i]); }
#endif
constexpr uint32_t kShadeStaticLds = 440u * 8u;
// a*b + c: one rounding where the FMA build of glibc fused it (kFused), two as the source is written otherwise
template <bool kFused>
MCRT_HD double fmaD(double a, double b, double c) {
    if (kFused) return __builtin_fma(a, b, c);  // (host emulation: correctly rounded with or without FMA hardware)
    return a * b + c;
}
MCRT_HD double absD(double v) { return bitsD(dBits(v) & 0x7FFFFFFFFFFFFFFFull); }
MCRT_HD double copySign(double mag, double sgn) { return bitsD((dBits(mag) & 0x7FFFFFFFFFFFFFFFull) | (dBits(sgn) & 0x8000000000000000ull)); }

// usncs.h / s_sin.c constants (bit patterns checked against the binary)
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ecep-7, kS3 = -0x1.a01a019db08b8p-13, kS4 = 0x1.71de27b9a7ed9p-19,
                 kS5 = -0x1.addffc2fcdf59p-26;
constexpr double kSn3 = -0x1.5555555555515p-3, kSn5 = 0x1.11110e829872fp-7, kCs2 = 0x1.0000000000000p-1, kCs4 = -0x1.5555555555535p-5,
                 kCs6 = 0x1.6c16bedd9e239p-10;
constexpr double kBig = 0x1.8000000000000p+45, kToInt = 0x1.8000000000000p+52, kHpInv = 0x1.45f306dc9c883p-1;
constexpr double kHp0 = 0x1.921fb54442d18p+0, kHp1 = 0x1.1a62633145c07p-54;
constexpr double kMp1 = 0x1.921fb58000000p+0, kMp2 = -0x1.dde973c000000p-27, kPp3 = -0x1.cb3b398000000p-55, kPp4 = -0x1.d747f23e32ed7p-83;

// TAYLOR_SIN(xx, x, dx): t = ((POLYNOMIAL(xx) * x - 0.5 * dx) * xx + dx); res = x + t
template <bool kFused>
MCRT_HD double taylorSin(double xx, double x, double dx) {
    double p = fmaD<kFused>(xx, kS5, kS4);  // s5 * xx + s4
    p = fmaD<kFused>(xx, p, kS3);
    p = fmaD<kFused>(xx, p, kS2);
    p = fmaD<kFused>(xx, p, kS1);           // POLYNOMIAL2(xx) + s1
    const double t = fmaD<kFused>(fmaD<kFused>(p, x, -(0.5 * dx)), xx, dx);
    return x + t;
}

// do_cos(x, dx)
template <bool kFused>
MCRT_HD double doCos(double x, double dx) {
    if (x < 0.0) dx = -dx;
    const double ax = absD(x);
    const double u = kBig + ax;
    x = (ax - (u - kBig)) + dx;
    const int k = (int)(unsigned)(dBits(u) & 0xFFFFFFFFull) * 4;
    const double xx = x * x;
    const double s = fmaD<kFused>(x * xx, fmaD<kFused>(xx, kSn5, kSn3), x);                // s = x + x * xx * (sn3 + xx * sn5)
    const double c = xx * fmaD<kFused>(xx, fmaD<kFused>(xx, kCs6, kCs4), kCs2);            // c = xx * (cs2 + xx * (cs4 + xx * cs6))
    const double sn = tabAt(k), ssn = tabAt(k + 1), cs = tabAt(k + 2), ccs = tabAt(k + 3);
    const double cor = fmaD<kFused>(-s, sn, fmaD<kFused>(-c, cs, fmaD<kFused>(-s, ssn, ccs)));     // cor = (ccs - s * ssn - cs * c) - sn * s
    return cs + cor;
}

// do_sin(x, dx)
template <bool kFused>
MCRT_HD double doSin(double x, double dx) {
    const double xold = x;
    if (absD(x) < 0.126) return taylorSin<kFused>(x * x, x, dx);
    if (x <= 0.0) dx = -dx;
    const double ax = absD(x);
    const double u = kBig + ax;
    x = ax - (u - kBig);
    const int k = (int)(unsigned)(dBits(u) & 0xFFFFFFFFull) * 4;
    const double xx = x * x;
    const double s = x + fmaD<kFused>(x * xx, fmaD<kFused>(xx, kSn5, kSn3), dx);           // s = x + (dx + x * xx * (sn3 + xx * sn5))
    const double c = fmaD<kFused>(x, dx, xx * fmaD<kFused>(xx, fmaD<kFused>(xx, kCs6, kCs4), kCs2));  // c = x * dx + xx * (cs2 + xx * (cs4 + xx * cs6))
    const double sn = tabAt(k), ssn = tabAt(k + 1), cs = tabAt(k + 2), ccs = tabAt(k + 3);
    const double cor = fmaD<kFused>(s, cs, fmaD<kFused>(-c, sn, fmaD<kFused>(s, ccs, ssn)));       // cor = (ssn + s * ccs - sn * c) + cs * s
    return copySign(sn + cor, xold);
}

// reduce_sincos(x, &a, &da): returns n (quadrant)
template <bool kFused>
MCRT_HD int reduceSinCos(double x, double& a, double& da) {
    const double t = fmaD<kFused>(x, kHpInv, kToInt);                              // t = x * hpinv + toint
    const double xn = t - kToInt;
    const double y = fmaD<kFused>(-xn, kMp2, fmaD<kFused>(-xn, kMp1, x));                  // y = (x - xn * mp1) - xn * mp2
    const int n = (int)(unsigned)(dBits(t) & 3ull);
    const double t2 = fmaD<kFused>(-xn, kPp3, y);                                  // t1 = xn * pp3; t2 = y - t1
    double db = fmaD<kFused>(-xn, kPp3, y - t2);                                   // db = (y - t2) - t1
    const double b = fmaD<kFused>(-xn, kPp4, t2);                                  // t1 = xn * pp4; b = t2 - t1
    db = db + fmaD<kFused>(-xn, kPp4, t2 - b);                                     // db += (t2 - b) - t1
    a = b;
    da = db;
    return n;
}

// do_sincos(a, da, n)
template <bool kFused>
MCRT_HD double doSinCos(double a, double da, int n) {
    const double r = (n & 1) ? doCos<kFused>(a, da) : doSin<kFused>(a, da);
    return (n & 2) ? -r : r;
}

}  // namespace glibc235

// __sin of the FMA build (s_sin.c:201-251)
MCRT_HD double refSin(double x) {
    using namespace glibc235;
    constexpr bool kFused = true;
    const unsigned k = (unsigned)(dBits(x) >> 32) & 0x7FFFFFFFu;
    if (k < 0x3e500000u) return x;                                                        // |x| < 2^-26
    if (k < 0x3feb6000u) return doSin<kFused>(x, 0.0);                                    // |x| < 0.855469
    if (k < 0x400368fdu) return copySign(doCos<kFused>(kHp0 - absD(x), kHp1), x);         // |x| < 2.426265
    if (k < 0x419921FBu) {                                                                // |x| < 105414350
        double a, da;
        const int n = reduceSinCos<kFused>(x, a, da);
        return doSinCos<kFused>(a, da, n);
    }
    return sin(x);
}

// __cos of the FMA build (s_sin.c:258-308)
MCRT_HD double refCos(double x) {
    using namespace glibc235;
    constexpr bool kFused = true;
    const unsigned k = (unsigned)(dBits(x) >> 32) & 0x7FFFFFFFu;
    if (k < 0x3e400000u) return 1.0;                                                      // |x| < 2^-27
    if (k < 0x3feb6000u) return doCos<kFused>(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = kHp0 - absD(x);
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        return doSin<kFused>(a, da);
    }
    if (k < 0x419921FBu) {
        double a, da;
        const int n = reduceSinCos<kFused>(x, a, da);
        return doSinCos<kFused>(a, da, n + 1);
    }
    return cos(x);
}

// __sincos (s_sincos.c:28-104), the baseline build: what the reference's sin / cos PAIRS compute. In every range it evaluates
// ONE do_sin and ONE do_cos of the same reduced argument (X, DX) - (x, 0) below 0.855469, (a, da) = hp0 - |x| in two pieces up
// to 2.426265, the Cody-Waite remainder beyond - and hands them out with the quadrant's signs; both kernels look up the same
// table entry. Written that way here (one table read, no branch repeated), every operation as in the source.
template <bool kLdsTab = true>
MCRT_HD void refSinCos(double x, double& sn_out, double& cs_out) {
    using namespace glibc235;
    constexpr bool kFused = false;
    const unsigned k = (unsigned)(dBits(x) >> 32) & 0x7FFFFFFFu;
    if (k < 0x3e400000u) {                                                                // |x| < 2^-27
        sn_out = x;
        cs_out = 1.0;
        return;
    }
    if (!(k < 0x419921FBu)) {                                                             // |x| >= 105414350, never reached by the path
        sincos(x, &sn_out, &cs_out);
        return;
    }
    double X, DX;
    bool swap, neg_s, neg_c, sin_takes_sign_of_x = false;  // sin = +-(swap ? do_cos : do_sin)(X, DX), cos = +-(swap ? do_sin : do_cos)(X, DX)
    if (k < 0x3feb6000u) {                                                                // |x| < 0.855469: sin = do_sin (x, 0), cos = do_cos (x, 0)
        X = x;
        DX = 0.0;
        swap = neg_s = neg_c = false;
    } else if (k < 0x400368fdu) {                                                         // |x| < 2.426265: sin = copysign (do_cos (a, da), x), cos = do_sin (a, da)
        const double y = kHp0 - absD(x);
        X = y + kHp1;
        DX = (y - X) + kHp1;
        swap = true;
        neg_s = neg_c = false;
        sin_takes_sign_of_x = true;
    } else {                                                                              // |x| < 105414350: do_sincos (a, da, n) and (a, da, n + 1)
        const int n = reduceSinCos<kFused>(x, X, DX);
        swap = (n & 1) != 0;
        neg_s = (n & 2) != 0;
        neg_c = ((n + 1) & 2) != 0;
    }
    // do_sin (X, DX) and do_cos (X, DX): same u, same table entry
    const double ax = absD(X);
    const double u = kBig + ax;
    const double xr = ax - (u - kBig);
    const int ti = (int)(unsigned)(dBits(u) & 0xFFFFFFFFull) * 4;
    const double sn = tabAt<kLdsTab>(ti), ssn = tabAt<kLdsTab>(ti + 1), cs = tabAt<kLdsTab>(ti + 2), ccs = tabAt<kLdsTab>(ti + 3);
    double S;
    if (ax < 0.126) {
        S = taylorSin<kFused>(X * X, X, DX);
    } else {
        const double dx = X <= 0.0 ? -DX : DX;
        const double xx = xr * xr;
        const double s = xr + (dx + xr * xx * (kSn3 + xx * kSn5));
        const double c = xr * dx + xx * (kCs2 + xx * (kCs4 + xx * kCs6));
        const double cor = (ssn + s * ccs - sn * c) + cs * s;
        S = copySign(sn + cor, X);
    }
    double C;
    {
        const double dx = X < 0.0 ? -DX : DX;
        const double xc = xr + dx;
        const double xx = xc * xc;
        const double s = xc + xc * xx * (kSn3 + xx * kSn5);
        const double c = xx * (kCs2 + xx * (kCs4 + xx * kCs6));
        const double cor = (ccs - s * ssn - cs * c) - sn * s;
        C = cs + cor;
    }
    double rs = swap ? C : S, rc = swap ? S : C;
    if (neg_s) rs = -rs;
    if (neg_c) rc = -rc;
    sn_out = sin_takes_sign_of_x ? copySign(rs, x) : rs;
    cs_out = rc;
}

// ---- asin as the reference computes it (Scene::skyColor, scene/scene.cpp:219-223: the one libm call of a sky scene's miss) ----
// glibc 2.35 __ieee754_asin (sysdeps/ieee754/dbl-64/e_asin.c, IBM Accurate Mathematical Library; the version without the
// multi-precision fall-backs): |x| < 2^-26: x; < 0.125: an odd polynomial; [0.125, 0.96875): per interval of width 2^-8 the Taylor
// expansion of asin around the interval's centre, coefficients from the table `asncs` (asincos.tbl); [0.96875, 1): pi/2 - 2 asin(sqrt
// ((1 - |x|) / 2)) with the root from a table seed, one Newton step and a double-double correction. std::asin is an IFUNC like sin
// (sysdeps/x86_64/fpu/multiarch/e_asin.c): on a CPU with FMA and AVX2 - the build container's and the GPU box's - the dynamic linker
// picks __ieee754_asin_fma, the same source compiled with -mfma. refAsin restates THAT instruction sequence (Ubuntu glibc
// 2.35-0ubuntu3.11, every a*b + c below is one vfmadd there); kFused = false evaluates the same expression tree with every operation
// rounded (not what any build of glibc computes bit for bit - its sse2 variant is compiled from the same source but this tree is the
// FMA build's; kept for the emulation's A/B counts, not used by the product). Tables:
// mcrt_glibc_asintab.inc (tools/make_glibc_asin_atan_tables.py). tests/test_libm.py: bit-equal to the host's asin on millions of arguments
// in every interval and across every interval boundary.
namespace glibc235 {
MCRT_LIBM_TABLE unsigned long long kAsinTab[2568 + 128] = {
#include "mcrt_glibc_asintab.inc"
};
MCRT_HD double asinTab(int i) { return bitsD(kAsinTab[i]); }
constexpr double kAf1 = 0x1.55555555554f9p-3, kAf2 = 0x1.333333336127dp-4, kAf3 = 0x1.6db6dae42c0e4p-5, kAf4 = 0x1.f1c7e04f4ad99p-6,
                 kAf5 = 0x1.6e442c822d419p-6, kAf6 = 0x1.292d80f453c72p-6;
constexpr double kRt0 = 0x1.fffffffecc1ddp-1, kRt1 = 0x1.fffffff757304p-2, kRt2 = 0x1.800496769c91ap-2, kRt3 = 0x1.4006318d1dab9p-2;
constexpr double kT24 = 16777216.0;

// one table interval: record at n, Taylor coefficients x[n+2 .. n+top], then x[n+top+1] (the low word of asin(x0)), the first-order
// term x[n+1] * xx, and x[n+top+2] (the high word) last
template <bool kFused>
MCRT_HD double asinInterval(double ax, int n, int top) {
    const double xx = ax - asinTab(n);
    double p = asinTab(n + top);
    for (int j = top - 1; j >= 2; j--) p = fmaD<kFused>(xx, p, asinTab(n + j));
    const double xx2 = xx * xx;
    p = fmaD<kFused>(xx2, p, asinTab(n + top + 1));
    const double r = fmaD<kFused>(xx, asinTab(n + 1), p);
    return r + asinTab(n + top + 2);
}
}  // namespace glibc235

template <bool kFused = true>
MCRT_HD double refAsinT(double x) {
    using namespace glibc235;
    const unsigned long long bits = dBits(x);
    const int m = (int)(uint32_t)(bits >> 32);
    const int k = m & 0x7fffffff;
    if (k < 0x3e500000) return x;                       // |x| < 2^-26
    if (k < 0x3fc00000) {                               // |x| < 0.125
        const double x2 = x * x;
        double p = fmaD<kFused>(x2, kAf6, kAf5);
        p = fmaD<kFused>(x2, p, kAf4);
        p = fmaD<kFused>(x2, p, kAf3);
        p = fmaD<kFused>(x2, p, kAf2);
        p = fmaD<kFused>(x2, p, kAf1);
        return fmaD<kFused>(p, x * x2, x);              // x + p * (x2 * x)
    }
    const double ax = m > 0 ? x : -x;
    double res;
    if (k < 0x3fe00000) {                               // [0.125, 0.5): 11-word records
        const int n = k < 0x3fd00000 ? 11 * ((k >> 15) & 0x1f) : 11 * ((k >> 14) & 0x3f) + 352;
        res = asinInterval<kFused>(ax, n, 6);
    } else if (k < 0x3fe80000) {                        // [0.5, 0.75): 12-word records
        res = asinInterval<kFused>(ax, 1056 + 3 * ((k >> 11) & 0x1fc), 7);
    } else if (k < 0x3fed8000) {                        // [0.75, 0.921875): 13-word records
        res = asinInterval<kFused>(ax, 992 + 13 * ((k >> 13) & 0x7f), 8);
    } else if (k < 0x3fee8000) {                        // [0.921875, 0.953125): 14-word records
        res = asinInterval<kFused>(ax, 884 + 14 * ((k >> 13) & 0x7f), 9);
    } else if (k < 0x3fef0000) {                        // [0.953125, 0.96875): 15-word records
        res = asinInterval<kFused>(ax, 768 + 15 * ((k >> 13) & 0x7f), 10);
    } else if (k < 0x3ff00000) {                        // [0.96875, 1)
        const double z = (1.0 - ax) * 0.5;
        const unsigned long long v = dBits(z);
        double t = asinTab(2568 + (int)((v >> 46) & 0x7f)) * bitsD((unsigned long long)(1023 + 0x1ff - (int)(v >> 53)) << 52);  // inroot[] * powtwo[]
        const double r = fmaD<kFused>(-(t * t), z, 1.0);
        double q = fmaD<kFused>(r, kRt3, kRt2);
        q = fmaD<kFused>(r, q, kRt1);
        q = fmaD<kFused>(r, q, kRt0);
        t = t * q;
        const double c = t * z;
        const double e = fmaD<kFused>(-c, t * 0.5, 1.5);
        const double y = (c + kT24) - kT24;
        const double tpy = fmaD<kFused>(e, c, y);
        const double cc = fmaD<kFused>(-y, y, z) / tpy;
        double p = fmaD<kFused>(z, kAf6, kAf5);
        p = fmaD<kFused>(z, p, kAf4);
        p = fmaD<kFused>(z, p, kAf3);
        p = fmaD<kFused>(z, p, kAf2);
        p = fmaD<kFused>(z, p, kAf1);
        p = p * z;
        const double ypc = y + cc;
        const double cor1 = fmaD<kFused>(-2.0, cc, kHp1);
        const double res1 = fmaD<kFused>(-2.0, y, kHp0);
        const double s2 = ypc + ypc;
        const double cor = fmaD<kFused>(-s2, p, cor1);
        res = cor + res1;
    } else if (k == 0x3ff00000 && (uint32_t)bits == 0u) {
        res = kHp0;                                     // |x| = 1
    } else if (k > 0x7ff00000 || (k == 0x7ff00000 && (uint32_t)bits != 0u)) {
        return x + x;                                   // NaN
    } else {
        const double d = x - x;                         // |x| > 1 (or infinite): invalid
        return d / d;
    }
    return m > 0 ? res : -res;
}
MCRT_HD double refAsin(double x) { return refAsinT<true>(x); }

// ---- atan2 as the reference computes it (Photon's constructor, integrator/photon-mapper/photon.hpp:10-11: theta and phi of a stored
// photon's direction are (float)std::atan2 (...)) ----
// glibc 2.35 __ieee754_atan2 (sysdeps/ieee754/dbl-64/e_atan2.c, IBM Accurate Mathematical Library, the version without the multi-
// precision fall-backs): u = min(|y|, |x|) / max(|y|, |x|) with the division's remainder du (EMULV: exact product through an fma),
// atan(u) by an odd polynomial below 1/16 or by the Taylor expansion around the nearest of 241 table points (cij, uatan.tbl), and
// the quadrant by adding to / subtracting from pi/2 or pi as double-doubles. An IFUNC like asin: restated is __ieee754_atan2_fma, the
// variant the build container's and the GPU box's CPUs select (every a*b + c below is one vfmadd there - including the table index
// (u * 256 + 2^52) - 2^52). Inputs with a NaN or an infinity go to the platform's atan2 (never produced by the path: the arguments are
// components of a unit vector). tests/test_libm.py: bit-equal to the host's atan2 on millions of argument pairs in every octant, on
// the axes, with extreme ratios and at the table-interval boundaries.
namespace glibc235 {
MCRT_LIBM_TABLE unsigned long long kAtanTab[241 * 7] = {
#include "mcrt_glibc_atantab.inc"
};
MCRT_HD double cij(int i, int j) { return bitsD(kAtanTab[7 * i + j]); }
constexpr double kD3 = -0x1.5555555555555p-2, kD5 = 0x1.99999999997fdp-3, kD7 = -0x1.24924923f7603p-3, kD9 = 0x1.c71c6e5129a3bp-4,
                 kD11 = -0x1.7458022b13c25p-4, kD13 = 0x1.375f08b31cbcep-4;
constexpr double kOpi = 0x1.921fb54442d18p+1, kOpi1 = 0x1.1a62633145c07p-53, kTwo52 = 0x1.0p+52, kTwo500 = 0x1.0p+500, kTwoM500 = 0x1.0p-500;

MCRT_HD double atanPoly(double v) {  // d3 + v (d5 + v (d7 + v (d9 + v (d11 + v d13))))
    double p = fmaD<true>(v, kD13, kD11);
    p = fmaD<true>(v, p, kD9);
    p = fmaD<true>(v, p, kD7);
    p = fmaD<true>(v, p, kD5);
    return fmaD<true>(v, p, kD3);
}
MCRT_HD int atanRow(double u) { return (int)(fmaD<true>(u, 256.0, kTwo52) - kTwo52) - 16; }
MCRT_HD double atanRowPoly(int i, double v) {  // cij[i][2] + v (cij[i][3] + v (cij[i][4] + v (cij[i][5] + v cij[i][6])))
    double q = fmaD<true>(v, cij(i, 6), cij(i, 5));
    q = fmaD<true>(v, q, cij(i, 4));
    q = fmaD<true>(v, q, cij(i, 3));
    return fmaD<true>(v, q, cij(i, 2));
}
}  // namespace glibc235

MCRT_HD double refAtan2(double y, double x) {
    using namespace glibc235;
    const unsigned long long by = dBits(y), bx = dBits(x);
    const uint32_t uy = (uint32_t)(by >> 32), ux = (uint32_t)(bx >> 32), dy = (uint32_t)by, dx = (uint32_t)bx;
    if ((ux & 0x7ff00000u) == 0x7ff00000u || (uy & 0x7ff00000u) == 0x7ff00000u) return atan2(y, x);  // NaN / infinity: not on the path
    if (uy == 0u && dy == 0u) return (ux & 0x80000000u) ? kOpi : 0.0;                                 // y = +0
    if (uy == 0x80000000u && dy == 0u) return (ux & 0x80000000u) ? -kOpi : -0.0;                      // y = -0
    if (x == 0.0) return (uy & 0x80000000u) ? -kHp0 : kHp0;
    double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
    if (de > 0x038fffff) return y > 0.0 ? kHp0 : -kHp0;                                               // |y| / |x| > 2^57
    if (de < -0x038fffff) {                                                                           // |y| / |x| < 2^-57
        if (x > 0.0) return copySign(ay / ax, y);
        return y > 0.0 ? kOpi : -kOpi;
    }
    if (ax < kTwoM500 || ay < kTwoM500) {
        ax *= kTwo500;
        ay *= kTwo500;
    }
    if (ax > kTwo500 || ay > kTwo500) {
        ax *= kTwoM500;
        ay *= kTwoM500;
    }
    double u, du, z;
    const bool y_smaller = ay < ax;
    {
        const double num = y_smaller ? ay : ax, den = y_smaller ? ax : ay;
        u = num / den;
        const double v = den * u, vv = fmaD<true>(den, u, -v);  // EMULV
        du = ((num - v) - vv) / den;
    }
    if (x > 0.0) {
        if (y_smaller) {                                   // (i) atan(ay / ax)
            if (u < 0.0625) {
                const double v = u * u;
                const double zz = fmaD<true>(u * v, atanPoly(v), du);
                z = u + zz;
            } else {
                const int i = atanRow(u);
                const double t3 = u - cij(i, 0);
                const double v = du + t3;                  // EADD (t3, du, v, dv)
                const double dv = absD(t3) > absD(du) ? (t3 - v) + du : (du - v) + t3;
                const double t2 = cij(i, 2);
                double q = fmaD<true>(v, cij(i, 6), cij(i, 5));
                q = fmaD<true>(v, q, cij(i, 4));
                q = fmaD<true>(v, q, cij(i, 3));
                q = (v * v) * q;
                q = fmaD<true>(dv, t2, q);
                const double zz = fmaD<true>(v, t2, q);
                z = zz + cij(i, 1);
            }
        } else {                                           // (ii) pi/2 - atan(ax / ay)
            if (u < 0.0625) {
                const double v = u * u;
                const double zz = (u * v) * atanPoly(v);
                const double t2 = kHp0 - u;                // ESUB (hpi, u, t2, cor)
                const double cor = kHp0 > absD(u) ? (kHp0 - t2) - u : kHp0 - (u + t2);
                const double t3 = ((cor + kHp1) - du) - zz;
                z = t3 + t2;
            } else {
                const int i = atanRow(u);
                const double v = (u - cij(i, 0)) + du;
                const double zz = fmaD<true>(-v, atanRowPoly(i, v), kHp1);
                z = (kHp0 - cij(i, 1)) + zz;
            }
        }
    } else if (ay > ax) {                                  // (iii) x < 0: pi/2 + atan(ax / ay)
        if (u < 0.0625) {
            const double v = u * u;
            const double zz = (v * u) * atanPoly(v);
            const double t2 = u + kHp0;                    // EADD (hpi, u, t2, cor)
            const double cor = kHp0 > absD(u) ? (kHp0 - t2) + u : (u - t2) + kHp0;
            const double t3 = ((cor + kHp1) + du) + zz;
            z = t3 + t2;
        } else {
            const int i = atanRow(u);
            const double v = (u - cij(i, 0)) + du;
            const double zz = fmaD<true>(v, atanRowPoly(i, v), kHp1);
            z = (kHp0 + cij(i, 1)) + zz;
        }
    } else {                                               // (iv) x < 0: pi - atan(ay / ax)
        if (u < 0.0625) {
            const double v = u * u;
            const double zz = (v * u) * atanPoly(v);
            const double t2 = kOpi - u;                    // ESUB (opi, u, t2, cor)
            const double cor = kOpi > absD(u) ? (kOpi - t2) - u : kOpi - (t2 + u);
            const double t3 = ((cor + kOpi1) - du) - zz;
            z = t3 + t2;
        } else {
            const int i = atanRow(u);
            const double v = (u - cij(i, 0)) + du;
            const double zz = fmaD<true>(-v, atanRowPoly(i, v), kOpi1);
            z = (kOpi - cij(i, 1)) + zz;
        }
    }
    return copySign(z, y);
}

// ---- sincosf: Photon::dir (photon.hpp:19-27) -----------------------------------------------------------------------------------------
// Photon::dir takes std::sin / std::cos of the photon's two FLOAT angles; g++ pairs them into sincosf calls (the reference binary
// imports sincosf), which on every x86-64 CPU with FMA + AVX2 resolves to __sincosf_fma: glibc 2.35's sysdeps/ieee754/flt-32/s_sincosf.c
// (Szabolcs Nagy's single-precision routines: the argument widened to double, one Cody-Waite step x - n pi/2 with n from x * 2^24 * 2/pi,
// two short polynomials in double, one rounding to float at the end) built with the x86 polynomial kernel sysdeps/x86/fpu/sincosf_poly.h
// and FMA contraction. The operation sequence below is that build's, read off its instructions (Ubuntu GLIBC 2.35-0ubuntu3.11,
// __sincosf_fma; every `a + b c` of the source is one fused multiply-add, the products x2 x, x2 x2, x3 x2, x4 x2 are plain): the
// coefficients are __sincosf_table's (libm's .rodata, checked by tests/test_libm.py against the running libm). Range: |y| < 120 - the
// photon angles are atan2 results, |y| <= pi; beyond, the platform's sincosf. tests/test_libm.py: EVERY float of [-4, 4] and a
// sample of the rest against the host's sincosf, sinf and cosf, bit for bit; on the GPU through mcrt_libm (MCRT_LIBM_SINCOSF).
MCRT_HD void refSinCosF(float y, float& sn_out, float& cs_out) {
    using glibc235::fmaD;
    constexpr double kFHpiInv = 0x1.45F306DC9C883p+23, kFHpi = 0x1.921FB54442D18p0;
    constexpr double kFC0 = 1.0, kFC1 = -0x1.ffffffd0c621cp-2, kFC2 = 0x1.55553e1068f19p-5, kFC3 = -0x1.6c087e89a359dp-10, kFC4 = 0x1.99343027bf8c3p-16;
    constexpr double kFS1 = -0x1.555545995a603p-3, kFS2 = 0x1.1107605230bc4p-7, kFS3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (floatBits(y) >> 20) & 0x7FFu;  // abstop12
    double x = (double)y, sign = 1.0, flip = 1.0;        // flip: the table for quadrants 2 and 3 is the first one's cosine coefficients negated
    int n = 0;
    if (top <= 0x3F3u) {                                 // |y| < pi/4
        if (top <= 0x397u) {                             // |y| < 2^-12
            sn_out = y;
            cs_out = 1.0f;
            return;
        }
    } else if (top <= 0x42Eu) {                          // |y| < 120: reduce_fast
        const double r = x * kFHpiInv;
        n = ((int)r + 0x800000) >> 24;                   // (cvttsd2si, arithmetic shift)
        x = fmaD<true>(-(double)n, kFHpi, x);             // vfnmadd: x - n * hpi, one rounding
        sign = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
        flip = (n & 2) ? -1.0 : 1.0;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        sn_out = sinf(y);
        cs_out = cosf(y);
#else
        sincosf(y, &sn_out, &cs_out);
#endif
        return;
    }
    const double xs = x * sign, x2 = x * x;
    const double x3 = x2 * xs, x4 = x2 * x2;
    const double s1 = fmaD<true>(x2, kFS3, kFS2), c2 = fmaD<true>(x2, flip * kFC4, flip * kFC3);
    const double x5 = x2 * x3, x6 = x2 * x4;
    const double c1 = fmaD<true>(x2, flip * kFC1, flip * kFC0);
    const double s = fmaD<true>(x3, kFS1, xs), c = fmaD<true>(x4, flip * kFC2, c1);
    const float sn = (float)fmaD<true>(s1, x5, s), cs = (float)fmaD<true>(c2, x6, c);
    sn_out = (n & 1) ? cs : sn;                          // odd quadrants swap the two
    cs_out = (n & 1) ? sn : cs;
}

}  // namespace mcrt
#endif  // MCRT_PLATFORM_LIBM
