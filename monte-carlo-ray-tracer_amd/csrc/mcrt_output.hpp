// Per-pixel arithmetic of Image::save (camera/image.cpp:37-88): the tone maps (camera/pixel-operators.cpp:7-44), sRGB gamma
// (color/srgb.hpp:55-63), byte truncation (pixel-operators.cpp:51-55) and the histogram bin of a brightness
// (common/histogram.cpp:21). Shared by the kernels of mcrt_output.hip and by the host build in tests/emu. glm's
// component-wise vector expressions are written out per component in glm's evaluation order; constants that the reference
// folds at compile time (C*B, D*E, D*F, E/F) are products/quotients of the same doubles here.
#pragma once

#include "../../include/mcrt.h"
#include "mcrt_libm_pow.hpp"
#include "mcrt_math.hpp"

namespace mcrt {

constexpr uint32_t kHistogramBins = 65536;  // Histogram(brightness, 65536), image.cpp:69,84

// filmicHable's inner f (pixel-operators.cpp:12-15): ((x*(A*x + C*B) + D*E) / (x*(A*x + B) + D*F)) - E/F
MCRT_HD double hableCurve(double x) {
    constexpr double A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

MCRT_HD d3 tonemapHable(d3 in) {  // pixel-operators.cpp:7-18
    const double w = hableCurve(11.2);
    return d3{hableCurve(in.x) / w, hableCurve(in.y) / w, hableCurve(in.z) / w};
}

MCRT_HD double clamp01(double x) {  // glm::clamp = min(max(x, 0), 1) with glm's comparisons
    const double lo = x < 0.0 ? 0.0 : x;
    return 1.0 < lo ? 1.0 : lo;
}

MCRT_HD d3 tonemapAces(d3 in) {  // pixel-operators.cpp:20-39; mat3 * vec3 = m[0][r]*x + m[1][r]*y + m[2][r]*z
    const d3 v{0.59719 * in.x + 0.35458 * in.y + 0.04823 * in.z,
               0.07600 * in.x + 0.90834 * in.y + 0.01566 * in.z,
               0.02840 * in.x + 0.13383 * in.y + 0.83777 * in.z};
    auto fit = [](double c) {
        const double a = c * (c + 0.0245786) - 0.000090537;
        const double b = c * (0.983729 * c + 0.4329510) + 0.238081;
        return a / b;
    };
    const d3 f{fit(v.x), fit(v.y), fit(v.z)};
    return d3{clamp01(1.60475 * f.x + -0.53108 * f.y + -0.07367 * f.z),
              clamp01(-0.10208 * f.x + 1.10813 * f.y + -0.00605 * f.z),
              clamp01(-0.00327 * f.x + -0.07276 * f.y + 1.07602 * f.z)};
}

// image.cpp:27-34: plain -> linear, "ACES" -> filmicACES, anything else -> filmicHable
MCRT_HD d3 tonemapApply(uint32_t tonemapper, bool plain, d3 in) {
    if (plain) return in;
    return tonemapper == MCRT_TONEMAP_ACES ? tonemapAces(in) : tonemapHable(in);
}

// glm::compAdd(v) / 3.0 (image.cpp:67,82; compAdd starts from T(0) and adds x, y, z in turn)
MCRT_HD double brightnessOf(d3 v) { return (((0.0 + v.x) + v.y) + v.z) / 3.0; }

// min((size_t)(v / bin_size), num_bins - 1), histogram.cpp:21 (v >= 0, bin_size > 0). A NaN lands in the last bin, as
// it does on x86-64 (cvttsd2si gives 2^63).
MCRT_HD uint32_t histogramBin(double v, double bin_size) {
    const double q = v / bin_size;
    return !(q < (double)(kHistogramBins - 1)) ? kHistogramBins - 1 : (uint32_t)q;
}

MCRT_HD double gammaCompress(double x) {  // srgb.hpp:55-63; std::pow = glibc's, restated (mcrt_libm_pow.hpp)
    return x <= 0.0031308 ? 12.92 * x : 1.055 * refPow(x, 1.0 / 2.4) - 0.055;
}

// truncate(sRGB::gammaCompress(tonemap(p * exposure) * gain)) -> bytes b, g, r (image.cpp:47, pixel-operators.cpp:51-55);
// nextafter(256.0, 0.0) = 256 - 2^-45
MCRT_HD void developPixel(uint32_t tonemapper, bool plain, d3 p, double exposure, double gain, uint8_t out[3]) {
    const d3 t = tonemapApply(tonemapper, plain, d3{p.x * exposure, p.y * exposure, p.z * exposure});
    const double scale = bitsD(0x406FFFFFFFFFFFFFull);
    const double r = clamp01(gammaCompress(t.x * gain)) * scale;
    const double g = clamp01(gammaCompress(t.y * gain)) * scale;
    const double b = clamp01(gammaCompress(t.z * gain)) * scale;
    out[0] = (uint8_t)b;
    out[1] = (uint8_t)g;
    out[2] = (uint8_t)r;
}

}  // namespace mcrt
