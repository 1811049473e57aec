// FP64 vector math for the gfx950 kernels, with the exact evaluation order of the reference's glm
// 0.9.9.8 calls (lib/glm/glm/detail/func_geometric.inl, func_common.inl), because per-pixel parity
// with the CPU reference depends on it: dot = (x*x' + y*y') + z*z', normalize(v) = v * (1/sqrt(dot)),
// min/max by compare-select (NaN behaviour of `(b < a) ? b : a`, not IEEE minNum).
// Built with -ffp-contract=off so that hipcc does not fuse a*b+c where g++ (x86-64, no -march) does not.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCRT_HD __host__ __device__ __forceinline__
#else
#define MCRT_HD inline
#endif

#if defined(__HIPCC__)
// Lane mask of a predicate. (HIP's __ballot(int) materialises 0 / 1 per lane and compares it with zero again: two extra
// vector instructions per ballot, and the wave-cooperative code lives on ballots.)
__device__ __forceinline__ unsigned long long waveBallot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif

// LDS (address space 3) pointers: telling the compiler that staged scene data lives in LDS turns the
// generic flat_load (which occupies the vector-memory path and both wait counters) into ds_read with
// 32-bit addressing. On the host build (tests/emu) the qualifier is empty.
#if defined(__HIP_DEVICE_COMPILE__)
#define MCRT_LDS_AS __attribute__((address_space(3)))
#else
#define MCRT_LDS_AS
#endif

// A kernel's dynamic LDS. (A macro so that the host emulation of workgroups, tests/emu/wave_emu.hpp, can give the kernels one array.)
#if !defined(MCRT_DYNAMIC_LDS)
#define MCRT_DYNAMIC_LDS(name, alignment) extern __shared__ __align__(alignment) unsigned char name[]
#endif

namespace mcrt {

template <class T, bool kLds>
struct PtrSel {
    using type = const T*;
};
template <class T>
struct PtrSel<T, true> {
    using type = MCRT_LDS_AS const T*;
};
template <class T, bool kLds>
using cptr = typename PtrSel<T, kLds>::type;

constexpr double kPi = 3.14159265358979323846;      // common/constants.hpp:5-8
constexpr double kInvPi = 0.31830988618379067154;
constexpr double kTwoPi = 6.283185307179586476925;
constexpr double kEpsilon = 1e-9;
constexpr double kDblMax = 1.7976931348623157e308;
constexpr uint32_t kNoSurface = 0xFFFFFFFFu;

struct d3 {
    double x, y, z;
};

MCRT_HD d3 mk(double x, double y, double z) { return d3{x, y, z}; }
MCRT_HD d3 splat(double s) { return d3{s, s, s}; }
MCRT_HD d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MCRT_HD d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MCRT_HD d3 operator*(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
MCRT_HD d3 operator/(d3 a, d3 b) { return d3{a.x / b.x, a.y / b.y, a.z / b.z}; }
MCRT_HD d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
MCRT_HD d3 operator*(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
MCRT_HD d3 operator/(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }
MCRT_HD d3 operator+(d3 a, double s) { return d3{a.x + s, a.y + s, a.z + s}; }
MCRT_HD d3 operator-(d3 a, double s) { return d3{a.x - s, a.y - s, a.z - s}; }
MCRT_HD d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
MCRT_HD d3 rcp3(d3 a) { return d3{1.0 / a.x, 1.0 / a.y, 1.0 / a.z}; }

MCRT_HD double sq(double x) { return x * x; }
MCRT_HD double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MCRT_HD d3 cross(d3 x, d3 y) {
    return d3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
MCRT_HD d3 normalize(d3 v) { return v * (1.0 / sqrt(dot(v, v))); }
MCRT_HD double gmin(double x, double y) { return (y < x) ? y : x; }  // glm::min / std::min(x,y)
MCRT_HD double gmax(double x, double y) { return (x < y) ? y : x; }  // glm::max / std::max(x,y)
// IEEE minNum/maxNum (one v_min_f64 / v_max_f64). Equal to gmin/gmax whenever neither input is NaN
// (up to the sign of a zero result, which no comparison can observe).
#if defined(__HIP_DEVICE_COMPILE__)
// (spelled as the instruction: through llvm.minnum the compiler adds a canonicalising v_max x,x per
// operand whenever the operands reach it through a phi)
MCRT_HD double fastMin(double x, double y) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
MCRT_HD double fastMax(double x, double y) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
#else
MCRT_HD double fastMin(double x, double y) { return fmin(x, y); }
MCRT_HD double fastMax(double x, double y) { return fmax(x, y); }
#endif
MCRT_HD unsigned long long dBits(double d) {
    union {
        double d;
        unsigned long long u;
    } c;
    c.d = d;
    return c.u;
}
MCRT_HD double bitsD(unsigned long long u) {
    union {
        double d;
        unsigned long long u;
    } c;
    c.u = u;
    return c.d;
}

MCRT_HD uint32_t floatBits(float f) {
    union {
        float f;
        uint32_t u;
    } c;
    c.f = f;
    return c.u;
}
MCRT_HD float bitsFloat(uint32_t u) {
    union {
        float f;
        uint32_t u;
    } c;
    c.u = u;
    return c.f;
}

MCRT_HD float floatAbove(double t) {  // smallest float >= t (t >= 0; +inf beyond the float range)
    float f = (float)t;
    if ((double)f < t) f = bitsFloat(floatBits(f) + 1u);
    return f;
}
MCRT_HD bool finite64(double x) { return fabs(x) <= kDblMax; }  // false for NaN and +-inf
MCRT_HD double compMax(d3 v) { return gmax(gmax(v.x, v.y), v.z); }
MCRT_HD double compMin(d3 v) { return gmin(gmin(v.x, v.y), v.z); }
MCRT_HD d3 mix(d3 x, d3 y, double a) { return x * (1.0 - a) + y * a; }
MCRT_HD double mix(double x, double y, double a) { return x * (1.0 - a) + y * a; }
MCRT_HD d3 sqrt3(d3 a) { return d3{sqrt(a.x), sqrt(a.y), sqrt(a.z)}; }
template <class P>
MCRT_HD d3 ld3(P p) {
    return d3{p[0], p[1], p[2]};
}

// glm::dmat3 by columns; only what CoordinateSystem needs (common/coordinate-system.cpp:7-35).
struct m3 {
    d3 c0, c1, c2;
};

// Duff et al. orthonormal basis, exactly as common/coordinate-system.cpp:7-18.
MCRT_HD m3 orthonormalBasis(d3 N) {
    double sign = copysign(1.0, N.z);
    double a = -1.0 / (sign + N.z);
    double b = N.x * N.y * a;
    return m3{d3{1.0 + sign * N.x * N.x * a, sign * b, -sign * N.x}, d3{b, sign + N.y * N.y * a, -N.y}, N};
}
// CoordinateSystem::from = T * v (lib/glm/glm/detail/type_mat3x3.inl:468-474)
MCRT_HD d3 csFrom(const m3& T, d3 v) {
    return d3{T.c0.x * v.x + T.c1.x * v.y + T.c2.x * v.z, T.c0.y * v.x + T.c1.y * v.y + T.c2.y * v.z,
              T.c0.z * v.x + T.c1.z * v.y + T.c2.z * v.z};
}
// CoordinateSystem::to = transpose(T) * v
MCRT_HD d3 csTo(const m3& T, d3 v) {
    return d3{T.c0.x * v.x + T.c0.y * v.y + T.c0.z * v.z, T.c1.x * v.x + T.c1.y * v.y + T.c1.z * v.z,
              T.c2.x * v.x + T.c2.y * v.y + T.c2.z * v.z};
}

}  // namespace mcrt
