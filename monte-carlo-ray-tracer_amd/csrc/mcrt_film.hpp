// Film reconstruction filters (camera/filter.hpp:10-64) and Film::deposit (camera/film.cpp:61-97): per-sample splats
// for every filter except the default box. Shared by the wavefront shade kernel and the CPU test harness.
#pragma once

#include "mcrt_math.hpp"
#include "mcrt_libm.hpp"
#include "../../include/mcrt.h"

namespace mcrt {

// MitchellNetravali<B, C>(x), filter.hpp:16-40 (x = 2 |t| / radius in [0, 2]); the coefficients are the constexpr
// expressions of the reference, evaluated in double like the compiler does.
MCRT_HD double mitchellNetravali(double B, double C, double x) {
    const double k = 6.0 / (6.0 - 2.0 * B);
    if (x < 1.0) {
        const double a = k * (12.0 - 9.0 * B - 6.0 * C) / 6.0;
        const double b = k * (-18.0 + 12.0 * B + 6.0 * C) / 6.0;
        const double d = k * (6.0 - 2.0 * B) / 6.0;
        return d + (b + a * x) * x * x;
    }
    const double a = k * (-B - 6.0 * C) / 6.0;
    const double b = k * (6.0 * B + 30.0 * C) / 6.0;
    const double c = k * (-12.0 * B - 48.0 * C) / 6.0;
    const double d = k * (8.0 * B + 24.0 * C) / 6.0;
    return d + (c + (b + a * x) * x) * x;
}

MCRT_HD double filmFilterFunction(uint32_t type, double x) {  // filter.hpp
    switch (type) {
        case MCRT_FILM_MITCHELL_NETRAVALI: return mitchellNetravali(1.0 / 3.0, 1.0 / 3.0, x);
        case MCRT_FILM_CATMULL_ROM: return mitchellNetravali(0.0, 0.5, x);
        case MCRT_FILM_B_SPLINE: return mitchellNetravali(1.0, 0.0, x);
        case MCRT_FILM_HERMITE: return mitchellNetravali(0.0, 0.0, x * 0.5);  // :52-56
        case MCRT_FILM_GAUSSIAN: {                                             // :58-63
            const double alpha = 2.0;
            return exp(-alpha * x * x) - exp(-alpha * 2.0 * 2.0);
        }
        case MCRT_FILM_LANCZOS:                                                // :65-69
            if (x == 0.0) return 1.0;
            return 2.0 * refSin(kPi * x) * refSin(kPi * x / 2.0) / (kPi * kPi * x * x);  // two sin calls of different arguments: glibc's sin (mcrt_libm.hpp)
        default: return 1.0;  // box
    }
}

inline double filmDefaultRadius(uint32_t type) {  // film.cpp:31-44
    switch (type) {
        case MCRT_FILM_MITCHELL_NETRAVALI: case MCRT_FILM_CATMULL_ROM: case MCRT_FILM_LANCZOS: return 2.0;
        case MCRT_FILM_B_SPLINE: return 1.39;
        case MCRT_FILM_HERMITE: return 1.0;
        case MCRT_FILM_GAUSSIAN: return 1.71;
        default: return 0.5;
    }
}

// A frame is SPLATTED (Film::deposit with a window, film.cpp:61-79) whenever the camera names a reconstruction filter - or keeps
// the box filter but gives it another radius than its default 0.5 (film.cpp:44-46: `set(Filter::box, 0.5)`, then
// `radius = getOptional(j, "radius", radius)`): every pixel within the radius then receives the sample with weight 1. Only the
// default box (a sample lands in its own pixel) takes the per-pixel sums. kFilmBoxSplat is FilmView::type of the widened box.
constexpr uint32_t kFilmBoxSplat = 0x100u;
MCRT_HD bool filmSplats(uint32_t film_filter, double film_radius) {
    return film_filter != MCRT_FILM_BOX || (film_radius != 0.0 && film_radius != 0.5);
}
MCRT_HD uint32_t filmViewType(uint32_t film_filter) { return film_filter == MCRT_FILM_BOX ? kFilmBoxSplat : film_filter; }

struct FilmView {
    uint32_t type;          // MCRT_FILM_*; MCRT_FILM_BOX = no splatting (per-pixel sums, film.cpp:13-17)
    uint32_t cache_size;
    double radius, two_inv_radius, inv_dx;
    const double* cache;    // [cache_size] filter_function(2 i / (cache_size - 1)), film.cpp:53-56
    double* blob;           // [height][width][4] rgb_sum, weight_sum (Film::Splat, film.hpp:33-43)
    uint32_t width, height;
};

MCRT_HD double filmFilter(const FilmView& f, double x) {  // Film::filter, film.cpp:86-97
    if (f.cache_size == 0) return filmFilterFunction(f.type, f.two_inv_radius * fabs(x));
    return f.cache[(size_t)(f.inv_dx * fabs(x) + 0.5)];
}

// Film::deposit, film.cpp:61-79. add(ptr, value) is the accumulation into a Splat component: an atomic add on the GPU
// (as the reference's std::atomic<double>), a plain one in the single-threaded harness.
template <class Add>
MCRT_HD void filmDeposit(const FilmView& f, double px, double py, d3 v, Add add) {
    long long min_x = (long long)(px + 0.5 - f.radius), min_y = (long long)(py + 0.5 - f.radius);   // ivec2(dvec2): truncation
    long long max_x = (long long)(px - 0.5 + f.radius), max_y = (long long)(py - 0.5 + f.radius);
    if (min_x < 0) min_x = 0;
    if (min_y < 0) min_y = 0;
    if (max_x > (long long)f.width - 1) max_x = (long long)f.width - 1;
    if (max_y > (long long)f.height - 1) max_y = (long long)f.height - 1;
    for (long long y = min_y; y <= max_y; y++) {
        const double weight_y = filmFilter(f, (double)y + 0.5 - py);
        for (long long x = min_x; x <= max_x; x++) {
            const double weight = weight_y * filmFilter(f, (double)x + 0.5 - px);
            double* s = f.blob + ((size_t)y * f.width + (size_t)x) * 4;
            add(s + 0, v.x * weight);  // Splat::update, film.cpp:99-105
            add(s + 1, v.y * weight);
            add(s + 2, v.z * weight);
            add(s + 3, weight);
        }
    }
}

// Splat::get, film.cpp:107-113
MCRT_HD void filmResolve(const double* splat, double* rgb) {
    const double w = splat[3];
    for (int c = 0; c < 3; c++) rgb[c] = w == 0.0 ? 0.0 : gmax(splat[c] / w, 0.0);
}

}  // namespace mcrt
