// Image::save on the GPU (camera/image.cpp:37-88): the frame stays where mcrt_render_device left it and 3 bytes per pixel
// leave the device instead of 24. Per pass p (0: exposure from the raw brightness, 1: gain from the tone-mapped one):
//   statKernel<p>   maximum brightness (bit pattern of a non-negative double orders like the integer) and "any negative"
//                   — Histogram's first loop (common/histogram.cpp:9-14)
//   histKernel<p>   65 536-bin histogram with bin_size = max / 65536 — its second loop (:16-22)
//   levelKernel     Histogram::level (:25-41) as one 1024-lane workgroup, then factor = target / L * 2^EV (image.cpp:39-40,
//                   70-71,85-86); the factor stays in device memory for the next pass
// then developKernel: truncate(gammaCompress(tonemap(p * exposure) * gain)) (image.cpp:47). No host round trip between the
// kernels. HBM-bound: the 24 B/pixel frame is read five times (120 B/pixel) and 3 B/pixel written — 0.25 GB for a
// 1920x1080 frame, some tens of microseconds per pass; the histogram atomics go to L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "mcrt_internal.hpp"
#include "mcrt_output.hpp"

using namespace mcrt;

namespace {

struct OutState {
    unsigned long long max_bits[2];
    uint32_t negative[2];
    double factor[2];  // exposure_factor, gain_factor
};

struct OutParams {
    const double* rgb;
    uint64_t pixels;
    uint32_t tonemapper;
    OutState* state;
    uint32_t* hist;  // [2][kHistogramBins]
};

template <int kPass>
__device__ double passBrightness(const OutParams& P, uint64_t i) {
    const d3 p{P.rgb[3 * i], P.rgb[3 * i + 1], P.rgb[3 * i + 2]};
    if (kPass == 0) return brightnessOf(p);
    const double e = P.state->factor[0];
    return brightnessOf(tonemapApply(P.tonemapper, false, d3{p.x * e, p.y * e, p.z * e}));
}

template <int kPass>
__global__ void __launch_bounds__(256) statKernel(OutParams P) {
    unsigned long long mx = 0;
    uint32_t neg = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.pixels; i += (uint64_t)gridDim.x * blockDim.x) {
        const double v = passBrightness<kPass>(P, i);
        if (v < 0.0) neg = 1;
        if (v > 0.0) {
            const unsigned long long b = dBits(v);
            mx = b > mx ? b : mx;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
        neg |= __shfl_xor(neg, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx) atomicMax(&P.state->max_bits[kPass], mx);
        if (neg) atomicOr(&P.state->negative[kPass], 1u);
    }
}

template <int kPass>
__global__ void __launch_bounds__(256) histKernel(OutParams P) {
    const double mx = bitsD(P.state->max_bits[kPass]);
    if (P.state->negative[kPass] || !(mx > 0.0)) return;  // no counts / every level 0 (see levelKernel)
    const double bin_size = mx / (double)kHistogramBins;
    uint32_t* hist = P.hist + kPass * kHistogramBins;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.pixels; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&hist[histogramBin(passBrightness<kPass>(P, i), bin_size)], 1u);
}

// Histogram::level(count_percentage): the first bin at which the running count reaches (size_t)(data_size * percentage).
// A negative brightness leaves the reference's histogram without bins (level 0); a frame whose maximum is 0 has
// bin_size 0, so its level is 0 whichever bin is found. L > 0 ? target / L : 1.0, times 2^EV.
__global__ void __launch_bounds__(1024) levelKernel(OutParams P, int pass, double percentage, double target, double scale) {
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t first;
    const uint32_t t = threadIdx.x;
    constexpr uint32_t kPer = kHistogramBins / 1024;
    const uint32_t* hist = P.hist + pass * kHistogramBins + t * kPer;
    unsigned long long sum = 0;
    for (uint32_t k = 0; k < kPer; k++) sum += hist[k];
    part[t] = sum;
    if (t == 0) first = kHistogramBins;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // inclusive scan
        const unsigned long long add = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    const unsigned long long num = (unsigned long long)((double)P.pixels * percentage);
    unsigned long long count = part[t] - sum;
    for (uint32_t k = 0; k < kPer; k++) {
        count += hist[k];
        if (count >= num) {
            atomicMin(&first, t * kPer + k);
            break;
        }
    }
    __syncthreads();
    if (t == 0) {
        const double mx = bitsD(P.state->max_bits[pass]);
        double level = 0.0;
        if (!P.state->negative[pass] && mx > 0.0 && first < kHistogramBins) level = (double)(first + 1) * (mx / (double)kHistogramBins);
        P.state->factor[pass] = (level > 0.0 ? target / level : 1.0) * scale;
    }
}

__global__ void __launch_bounds__(256) developKernel(OutParams P, int plain, uint8_t* bgr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.pixels) return;
    const d3 p{P.rgb[3 * i], P.rgb[3 * i + 1], P.rgb[3 * i + 2]};
    uint8_t out[3];
    developPixel(P.tonemapper, plain != 0, p, plain ? 1.0 : P.state->factor[0], plain ? 1.0 : P.state->factor[1], out);
    bgr[3 * i] = out[0];
    bgr[3 * i + 1] = out[1];
    bgr[3 * i + 2] = out[2];
}

#define OUT_TRY(call)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            if (scratch) (void)hipFree(scratch);                                                                  \
            return ctxFail(ctx, MCRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                 \
        }                                                                                                         \
    } while (0)

}  // namespace

extern "C" int mcrt_tonemap_device(mcrt_ctx* ctx, const double* d_rgb, const mcrt_image_desc* image, uint8_t* d_bgr, double* factors,
                                   void* stream_) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!d_rgb || !image || !d_bgr || image->width == 0 || image->height == 0 || image->tonemapper > MCRT_TONEMAP_ACES)
        return ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_tonemap_device: bad argument");
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : (hipStream_t)ctxStream(ctx);
    void* scratch = nullptr;
    OUT_TRY(hipSetDevice(ctxDevice(ctx)));
    const size_t hist_bytes = 2 * (size_t)kHistogramBins * sizeof(uint32_t);
    OUT_TRY(hipMalloc(&scratch, sizeof(OutState) + hist_bytes));
    OUT_TRY(hipMemsetAsync(scratch, 0, sizeof(OutState) + hist_bytes, stream));
    OutParams P;
    P.rgb = d_rgb;
    P.pixels = (uint64_t)image->width * image->height;
    P.tonemapper = image->tonemapper;
    P.state = (OutState*)scratch;
    P.hist = (uint32_t*)((char*)scratch + sizeof(OutState));
    const uint32_t per_pixel = (uint32_t)((P.pixels + 255) / 256);
    const uint32_t strided = per_pixel < 2048u ? per_pixel : 2048u;  // 8 workgroups per CU, grid-stride
    if (!image->plain) {
        // Image::Image (image.cpp:19-23): std::pow(2, EV)
        const double exposure_scale = std::pow(2.0, image->exposure_compensation), gain_scale = std::pow(2.0, image->gain_compensation);
        statKernel<0><<<strided, 256, 0, stream>>>(P);
        histKernel<0><<<strided, 256, 0, stream>>>(P);
        levelKernel<<<1, 1024, 0, stream>>>(P, 0, 0.5, 0.5, exposure_scale);
        statKernel<1><<<strided, 256, 0, stream>>>(P);
        histKernel<1><<<strided, 256, 0, stream>>>(P);
        levelKernel<<<1, 1024, 0, stream>>>(P, 1, 0.99, 0.99, gain_scale);
    }
    developKernel<<<per_pixel, 256, 0, stream>>>(P, (int)image->plain, d_bgr);
    OUT_TRY(hipGetLastError());
    OutState st;
    OUT_TRY(hipMemcpyAsync(&st, scratch, sizeof st, hipMemcpyDeviceToHost, stream));
    OUT_TRY(hipStreamSynchronize(stream));
    if (factors) {
        factors[0] = image->plain ? 1.0 : st.factor[0];
        factors[1] = image->plain ? 1.0 : st.factor[1];
    }
    (void)hipFree(scratch);
    return MCRT_OK;
}

extern "C" int mcrt_tonemap(mcrt_ctx* ctx, const double* rgb, const mcrt_image_desc* image, uint8_t* bgr, double* factors) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!rgb || !image || !bgr || image->width == 0 || image->height == 0) return ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_tonemap: bad argument");
    const size_t pixels = (size_t)image->width * image->height;
    void* scratch = nullptr;
    OUT_TRY(hipSetDevice(ctxDevice(ctx)));
    OUT_TRY(hipMalloc(&scratch, pixels * 27));
    double* d_rgb = (double*)scratch;
    uint8_t* d_bgr = (uint8_t*)scratch + pixels * 24;
    hipStream_t stream = (hipStream_t)ctxStream(ctx);
    OUT_TRY(hipMemcpyAsync(d_rgb, rgb, pixels * 24, hipMemcpyHostToDevice, stream));
    const int rc = mcrt_tonemap_device(ctx, d_rgb, image, d_bgr, factors, stream);
    if (rc != MCRT_OK) {
        (void)hipFree(scratch);
        return rc;
    }
    OUT_TRY(hipMemcpyAsync(bgr, d_bgr, pixels * 3, hipMemcpyDeviceToHost, stream));
    OUT_TRY(hipStreamSynchronize(stream));
    (void)hipFree(scratch);
    return MCRT_OK;
}

// mcrt_libm(MCRT_LIBM_POW): the pow the develop kernel inlines, on arrays (known-answer test against the host's glibc, tests/test_libm.py)
namespace {
__global__ void __launch_bounds__(256) powKatKernel(uint64_t n, const double* a, const double* b, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = mcrt::refPow(a[i], b[i]);
}
}  // namespace
int mcrt::launchPowKat(void* stream, uint64_t n, const double* a, const double* b, double* out) {
    const uint32_t grid = (uint32_t)std::min<uint64_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(powKatKernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, a, b, out);
    return (int)hipGetLastError();
}

// HeaderTGA (camera/image.hpp:39-50): 12 bytes {0, 0, 2, 0...}, width, height (little-endian 16 bit), {24, 32}; then the
// pixels left to right, top to bottom, 3 bytes each (image.cpp:42-51).
extern "C" int mcrt_tga_save(const char* path, uint32_t width, uint32_t height, const uint8_t* bgr) {
    if (!path || !bgr || width == 0 || height == 0 || width > 0xFFFFu || height > 0xFFFFu) return MCRT_ERR_INVALID;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return MCRT_ERR_IO;
    const uint8_t header[18] = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint8_t)(width & 0xFF), (uint8_t)(width >> 8),
                                (uint8_t)(height & 0xFF), (uint8_t)(height >> 8), 24, 32};
    const size_t bytes = (size_t)width * height * 3;
    const bool ok = std::fwrite(header, 1, sizeof header, f) == sizeof header && std::fwrite(bgr, 1, bytes, f) == bytes;
    return (std::fclose(f) == 0 && ok) ? MCRT_OK : MCRT_ERR_IO;
}
