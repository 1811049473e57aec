// Known-byte access patterns for calibrating rocprofv3's L2 memory-side counters on gfx950 (MI355X_MICROARCH.md, HBM
// section: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// One kernel per pattern of the integrator kernels:
//   calibStream16   16 B per lane, consecutive            (photon sort / sample resolve; the guide's calibrated case)
//   calibBlock64    64-byte blocks at random, 4 lanes x 16 B per block   (quantised child blocks, mcrt_qbvh.hpp)
//   calibRecord64   64-byte records at random, ONE lane reads all four 16-byte pieces (nodes64 of the lane state machine)
//   calibRun8k      8 KB runs at random, a wave reads 256 photons of 32 B as 2 x 16 B per lane (leaf scans, mcrt_waveknn.hpp)
//   calibPrim80     80-byte primitive records at random, one lane reads 5 x 16 B (leaf steps)
//   calibWrite8     8 B per lane, consecutive stores       (pool planes of the wavefront pipeline)
//   calibWrite24    24 B per lane at random 24-byte slots  (per-sample store written by finished paths)
// Usage: traffic_calib <working set MiB> <MiB to move per pattern>   -> one JSON line with the bytes each kernel asked for.
// Run under `rocprofv3 --kernel-trace --pmc ...` by tools/calibrate_traffic.py.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            std::exit(2);                                                                       \
        }                                                                                       \
    } while (0)

namespace {

__device__ inline uint64_t mix(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ inline void fold(uint4 v, uint32_t& acc) { acc ^= v.x ^ v.y ^ v.z ^ v.w; }

__global__ void calibStream16(const uint4* src, uint64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) fold(src[i], acc);
    if (acc == 0x12345u) sink[0] = acc;
}

// iters blocks per 4-lane group; nblocks = working set / 64
__global__ void calibBlock64(const uint4* src, uint64_t nblocks, uint32_t iters, uint32_t* sink) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t group = gid >> 2;
    uint32_t piece = (uint32_t)gid & 3u, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t b = mix(group * 0x100000001B3ull + it) % nblocks;
        fold(src[b * 4 + piece], acc);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ void calibRecord64(const uint4* src, uint64_t nblocks, uint32_t iters, uint32_t* sink) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t b = mix(gid * 0x100000001B3ull + it) % nblocks;
        const uint4* p = src + b * 4;
        uint4 a = p[0], c = p[1], d = p[2], e = p[3];
        fold(a, acc); fold(c, acc); fold(d, acc); fold(e, acc);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ void calibPrim80(const uint4* src, uint64_t nrec, uint32_t iters, uint32_t* sink) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t b = mix(gid * 0x100000001B3ull + it) % nrec;
        const uint4* p = src + b * 5;
        uint4 a = p[0], c = p[1], d = p[2], e = p[3], f = p[4];
        fold(a, acc); fold(c, acc); fold(d, acc); fold(e, acc); fold(f, acc);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// a wave reads one 8 KB run (256 records of 32 B) per iteration: 4 round trips of 64 records, 2 x 16 B per lane
__global__ void calibRun8k(const uint4* src, uint64_t nruns, uint32_t iters, uint32_t* sink) {
    uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t r = mix(wave * 0x100000001B3ull + it) % nruns;
        const uint4* p = src + r * 512;  // 8 KB = 512 x 16 B
        for (uint32_t q = 0; q < 4; q++) {
            uint4 a = p[(q * 64 + lane) * 2], b = p[(q * 64 + lane) * 2 + 1];
            fold(a, acc); fold(b, acc);
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ void calibWrite8(uint64_t* dst, uint64_t n8) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = i;
}

__global__ void calibWrite24(double* dst, uint64_t nslots, uint32_t iters) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < iters; it++) {
        uint64_t s = mix(gid * 0x100000001B3ull + it) % nslots;
        dst[s * 3] = (double)it; dst[s * 3 + 1] = 1.0; dst[s * 3 + 2] = 2.0;
    }
}

}  // namespace

int main(int argc, char** argv) {
    uint64_t ws = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096) << 20;
    uint64_t move = (argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 16384) << 20;
    void* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, ws));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, ws));
    CHECK(hipDeviceSynchronize());
    const uint32_t block = 256, grid = 256 * 8 * 4;  // 8192 workgroups of 4 waves
    const uint64_t lanes = (uint64_t)block * grid;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("{\"working_set_bytes\": %llu", (unsigned long long)ws);
    auto timed = [&](const char* name, double bytes, auto&& launch) {
        launch();  // warm (page tables, clocks)
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // two launches per pattern: the counters of a kernel name are summed over both, "bytes" is per launch
        std::printf(", \"%s\": {\"bytes_per_launch\": %.0f, \"launches\": 2, \"ms\": %.4f, \"GBs\": %.1f}", name, bytes, ms, bytes / ms / 1e6);
    };
    {
        uint64_t n16 = (move < ws ? move : ws) / 16;
        timed("calibStream16", (double)n16 * 16, [&] { hipLaunchKernelGGL(calibStream16, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, n16, sink); });
    }
    {
        uint64_t groups = lanes / 4;
        uint32_t iters = (uint32_t)(move / 64 / groups);
        if (iters < 1) iters = 1;
        timed("calibBlock64", (double)groups * iters * 64, [&] { hipLaunchKernelGGL(calibBlock64, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, ws / 64, iters, sink); });
    }
    {
        uint32_t iters = (uint32_t)(move / 64 / lanes);
        if (iters < 1) iters = 1;
        timed("calibRecord64", (double)lanes * iters * 64, [&] { hipLaunchKernelGGL(calibRecord64, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, ws / 64, iters, sink); });
    }
    {
        uint32_t iters = (uint32_t)(move / 80 / lanes);
        if (iters < 1) iters = 1;
        timed("calibPrim80", (double)lanes * iters * 80, [&] { hipLaunchKernelGGL(calibPrim80, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, ws / 80, iters, sink); });
    }
    {
        uint64_t waves = lanes / 64;
        uint32_t iters = (uint32_t)(move / 8192 / waves);
        if (iters < 1) iters = 1;
        timed("calibRun8k", (double)waves * iters * 8192, [&] { hipLaunchKernelGGL(calibRun8k, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, ws / 8192, iters, sink); });
    }
    {
        uint64_t n8 = (move < ws ? move : ws) / 8;
        timed("calibWrite8", (double)n8 * 8, [&] { hipLaunchKernelGGL(calibWrite8, dim3(grid), dim3(block), 0, 0, (uint64_t*)buf, n8); });
    }
    {
        uint32_t iters = (uint32_t)(move / 4 / 24 / lanes);  // a quarter of the volume: scattered partial-line writes are slow
        if (iters < 1) iters = 1;
        timed("calibWrite24", (double)lanes * iters * 24, [&] { hipLaunchKernelGGL(calibWrite24, dim3(grid), dim3(block), 0, 0, (double*)buf, ws / 24, iters); });
    }
    std::printf("}\n");
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
