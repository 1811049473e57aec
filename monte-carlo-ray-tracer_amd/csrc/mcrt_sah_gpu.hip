// mcrt_sah_gpu.hip — the reference's binned-SAH hierarchies (bvh/bvh.cpp:165-426) built level by level on the GPU.
// Algorithm and why the tree is the reference's: mcrt_sah_shared.hpp. Per level, over ALL open nodes at once:
//   1. segment of every position of the working order (binary search over the open nodes' starts)
//   2. centroid bounds per open node: atomic min / max on order-preserving 64-bit images of the doubles (exact)
//   3. plan per open node (one thread each): rule, axes, or "no usable axis"
//   4. bin of every surface; per (node, bin) an atomic count and an atomic box
//   5. split per open node (one thread each, the reference's cost loop over the bins — tiny next to the passes)
//   6. order-preserving partition: child of every surface, exclusive prefix sums over the four children's indicators (two
//      64-bit words of two 32-bit counters each), new position = run start + child's offset + own rank; the parts of an
//      arbitrary split are dealt in closed form; surfaces of closed nodes stay where they are
//   7. (rare) boxes of round-robin parts: one more atomic pass over those runs
// The host reads back the splits (a few hundred bytes per open node), appends the children to the node table and uploads
// the next level's open nodes; the depth-first numbering (BVH::compact) is a host pass over the finished table.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_internal.hpp"
#include "mcrt_sah_shared.hpp"

using namespace mcrt;

namespace {

struct Dev {
    void* p = nullptr;
    size_t bytes = 0;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t want) {  // grow-only
        if (want <= bytes && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const hipError_t e = hipMalloc(&p, want ? want : 1);
        if (e == hipSuccess) bytes = want ? want : 1;
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// doubles as unsigned integers of the same order (-0.0 below +0.0; no NaNs among box coordinates)
__device__ inline unsigned long long orderedKey(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double keyDouble(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}
constexpr unsigned long long kKeyOfMax = 0xFFEFFFFFFFFFFFFFull;  // orderedKey(+DBL_MAX): the empty box's minimum
constexpr unsigned long long kKeyOfMin = 0x0010000000000000ull;  // orderedKey(-DBL_MAX): the empty box's maximum

constexpr uint32_t kNoSeg = 0xFFFFFFFFu;

__global__ void surfaceBoxKernel(const uint8_t* kind, const double* surf_v, const double* quadrics, uint64_t n, double* bb, double* centroid,
                                 uint32_t* order) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double b[6];
    surfaceBounds(kind[i], surf_v + 9 * i, quadrics, b);
    for (int c = 0; c < 6; c++) bb[i * 6 + c] = b[c];
    for (int c = 0; c < 3; c++) centroid[i * 3 + c] = (b[3 + c] + b[c]) / 2.0;  // BB().centroid()
    order[i] = (uint32_t)i;
}

// position -> open node (the open nodes are sorted by start and disjoint)
__global__ void segmentOfKernel(const SahSeg* segs, uint32_t S, uint64_t n, uint32_t* seg_of) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t lo = 0, hi = S;  // the last segment with start <= p
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].start <= p) lo = mid + 1;
        else hi = mid;
    }
    uint32_t s = kNoSeg;
    if (lo > 0 && p < (uint64_t)segs[lo - 1].start + segs[lo - 1].size) s = lo - 1;
    seg_of[p] = s;
}

__global__ void fillKeysKernel(unsigned long long* keys, uint64_t boxes) {  // [boxes][6] <- the empty box
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= boxes * 6) return;
    keys[i] = (i % 6) < 3 ? kKeyOfMax : kKeyOfMin;
}

// (Near the root every surface belongs to one of a few open nodes: 491 593 atomics on the same six words took 40 ms. The
// lanes of a wave hold consecutive positions, so most waves lie inside ONE node: they reduce in registers and send six
// atomics; a wave that straddles nodes sends its lanes' own.)
__global__ void centroidBoundsKernel(const uint32_t* order, const uint32_t* seg_of, const double* centroid, uint64_t n, unsigned long long* ce) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = p < n ? seg_of[p] : kNoSeg;
    unsigned long long lo[3] = {kKeyOfMax, kKeyOfMax, kKeyOfMax}, hi[3] = {kKeyOfMin, kKeyOfMin, kKeyOfMin};
    if (s != kNoSeg) {
        const double* c = centroid + (size_t)order[p] * 3;
        for (int k = 0; k < 3; k++) lo[k] = hi[k] = orderedKey(c[k]);
    }
    const unsigned long long any = __ballot(s != kNoSeg);
    if (!any) return;
    const uint32_t s0 = (uint32_t)__shfl((int)s, __ffsll((long long)any) - 1, 64);  // the node of the first lane that has one
    const unsigned long long same = __ballot(s == s0 || s == kNoSeg);
    if (same == __ballot(true)) {  // one open node (lanes past the end or in closed runs hold neutral values)
        for (int k = 0; k < 3; k++)
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long a = __shfl_xor(lo[k], off, 64), b = __shfl_xor(hi[k], off, 64);
                lo[k] = a < lo[k] ? a : lo[k];
                hi[k] = b > hi[k] ? b : hi[k];
            }
        if ((threadIdx.x & 63u) == 0u)
            for (int k = 0; k < 3; k++) {
                atomicMin(ce + (size_t)s0 * 6 + k, lo[k]);
                atomicMax(ce + (size_t)s0 * 6 + 3 + k, hi[k]);
            }
        return;
    }
    if (s == kNoSeg) return;
    for (int k = 0; k < 3; k++) {
        atomicMin(ce + (size_t)s * 6 + k, lo[k]);
        atomicMax(ce + (size_t)s * 6 + 3 + k, hi[k]);
    }
}

__global__ void planKernel(const SahSeg* segs, uint32_t S, const unsigned long long* ce, SahPlan* plans) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double b[6];
    for (int k = 0; k < 6; k++) b[k] = keyDouble(ce[(size_t)s * 6 + k]);
    sahPlan(segs[s], b, plans[s]);
}

__global__ void binKernel(const uint32_t* order, const uint32_t* seg_of, const double* centroid, const double* bb, uint64_t n, const SahPlan* plans,
                          int bins, uint32_t cells, uint32_t* cell_of, uint32_t* count, unsigned long long* bbox) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = seg_of[p];
    if (s == kNoSeg) return;
    const SahPlan& pl = plans[s];
    if (pl.mode != kSahBinary && pl.mode != kSahQuad) return;
    const uint32_t surf = order[p];
    const uint32_t c = sahCell(pl, centroid + (size_t)surf * 3, bins);
    cell_of[p] = c;
    atomicAdd(count + (size_t)s * cells + c, 1u);
    unsigned long long* box = bbox + ((size_t)s * cells + c) * 6;
    const double* b = bb + (size_t)surf * 6;
    for (int k = 0; k < 3; k++) {
        atomicMin(box + k, orderedKey(b[k]));
        atomicMax(box + 3 + k, orderedKey(b[3 + k]));
    }
}

__global__ void decodeKeysKernel(unsigned long long* keys, uint64_t count) {  // in place: ordered keys -> the doubles' bits
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    keys[i] = (unsigned long long)__double_as_longlong(keyDouble(keys[i]));
}

__global__ void splitKernel(const SahSeg* segs, uint32_t S, const SahPlan* plans, int bins, uint32_t cells, const uint32_t* count, const double* bbox,
                            SahSplit* splits) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    sahEvaluate(segs[s], plans[s], bins, count + (size_t)s * cells, bbox + (size_t)s * cells * 6, splits[s]);
}

// the four children's indicators of a position, packed: word 0 = {child 0, child 1}, word 1 = {child 2, child 3} (32 bits each)
__global__ void childFlagsKernel(const uint32_t* seg_of, const uint32_t* cell_of, uint64_t n, const SahSplit* splits, int bins, uint8_t* child_of,
                                 unsigned long long* flags01, unsigned long long* flags23) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = seg_of[p];
    unsigned long long f01 = 0ull, f23 = 0ull;
    uint8_t child = 0xFF;
    if (s != kNoSeg && (splits[s].mode == kSahBinary || splits[s].mode == kSahQuad)) {
        child = (uint8_t)sahChildOfCell(splits[s], cell_of[p], bins);
        if (child < 2) f01 = 1ull << (32 * child);
        else f23 = 1ull << (32 * (child - 2));
    }
    child_of[p] = child;
    flags01[p] = f01;
    flags23[p] = f23;
}

__global__ void scatterKernel(const uint32_t* order, const uint32_t* seg_of, const uint8_t* child_of, uint64_t n, const SahSeg* segs, const SahSplit* splits,
                              const unsigned long long* scan01, const unsigned long long* scan23, uint32_t* order_out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = seg_of[p];
    uint64_t to = p;  // closed nodes and nodes that stay leaves: in place
    if (s != kNoSeg) {
        const SahSeg& g = segs[s];
        const SahSplit& sp = splits[s];
        if (sp.mode == kSahBinary || sp.mode == kSahQuad) {
            const uint32_t c = child_of[p];
            uint32_t offset = 0;
            for (uint32_t k = 0; k < c; k++) offset += sp.child_size[k];
            const unsigned long long here = c < 2 ? scan01[p] : scan23[p], first = c < 2 ? scan01[g.start] : scan23[g.start];
            const uint32_t rank = (uint32_t)((here >> (32 * (c & 1u))) & 0xFFFFFFFFull) - (uint32_t)((first >> (32 * (c & 1u))) & 0xFFFFFFFFull);
            to = (uint64_t)g.start + offset + rank;
        } else if (sp.mode == kSahArb) {  // arbitrarySplit, bvh.cpp:451-473: surface i of the node -> part i % N, order kept
            const uint32_t i = (uint32_t)(p - g.start), c = i % sp.arb;
            uint32_t offset = 0;
            for (uint32_t k = 0; k < c; k++) offset += sp.child_size[k];
            to = (uint64_t)g.start + offset + i / sp.arb;
        }
    }
    order_out[to] = order[p];
}

struct Run { uint32_t start, size; };
__global__ void runBoxKernel(const Run* runs, const uint32_t* order, const double* bb, unsigned long long* out) {  // one block per run
    const Run r = runs[blockIdx.x];
    unsigned long long* box = out + (size_t)blockIdx.x * 6;
    for (uint32_t i = threadIdx.x; i < r.size; i += blockDim.x) {
        const double* b = bb + (size_t)order[r.start + i] * 6;
        for (int k = 0; k < 3; k++) {
            atomicMin(box + k, orderedKey(b[k]));
            atomicMax(box + 3 + k, orderedKey(b[3 + k]));
        }
    }
}

#define SAH_TRY(call)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return ctxFail(ctx, MCRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline uint32_t gridFor(uint64_t n, uint32_t block = 256) { return (uint32_t)((n + block - 1) / block); }

}  // namespace

int mcrt::bvhSahGpu(mcrt_ctx* ctx, const mcrt_scene_desc* s, int arity, int bins, mcrt_bvh* B) {
    const bool timing = ctxOptOn(ctx, "MCRT_SAH_TIME");  // phase times to stderr
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    SAH_TRY(hipSetDevice(ctxDevice(ctx)));
    const uint64_t n = s->num_surfaces;
    if (n > 0x7FFFFFFFull) return ctxFail(ctx, MCRT_ERR_UNSUPPORTED, "more than 2^31-1 surfaces (prefix sum item count)");
    const uint32_t cells = (uint32_t)(bins * bins);
    Dev d_kind, d_v, d_q, d_bb, d_centroid, d_order, d_order2, d_seg_of, d_cell_of, d_child_of, d_f01, d_f23, d_s01, d_s23, d_tmp;
    Dev d_segs, d_ce, d_plans, d_count, d_bbox, d_splits, d_runs, d_run_box;
    SAH_TRY(d_kind.reserve(n));
    SAH_TRY(d_v.reserve(n * 72));
    SAH_TRY(d_q.reserve((size_t)s->num_quadrics * 22 * 8));
    SAH_TRY(d_bb.reserve(n * 48));
    SAH_TRY(d_centroid.reserve(n * 24));
    SAH_TRY(d_order.reserve(n * 4));
    SAH_TRY(d_order2.reserve(n * 4));
    SAH_TRY(d_seg_of.reserve(n * 4));
    SAH_TRY(d_cell_of.reserve(n * 4));
    SAH_TRY(d_child_of.reserve(n));
    SAH_TRY(d_f01.reserve(n * 8));
    SAH_TRY(d_f23.reserve(n * 8));
    SAH_TRY(d_s01.reserve(n * 8));
    SAH_TRY(d_s23.reserve(n * 8));
    SAH_TRY(hipMemcpy(d_kind.p, s->surf_kind, n, hipMemcpyHostToDevice));
    SAH_TRY(hipMemcpy(d_v.p, s->surf_v, n * 72, hipMemcpyHostToDevice));
    if (s->num_quadrics) SAH_TRY(hipMemcpy(d_q.p, s->quadrics, (size_t)s->num_quadrics * 22 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(surfaceBoxKernel, dim3(gridFor(n)), dim3(256), 0, 0, d_kind.as<uint8_t>(), d_v.as<double>(), d_q.as<double>(), n,
                       d_bb.as<double>(), d_centroid.as<double>(), d_order.as<uint32_t>());
    SAH_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    SAH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_f01.as<unsigned long long>(), d_s01.as<unsigned long long>(), (int)n));
    SAH_TRY(d_tmp.reserve(tmp_bytes));
    {   // an open node holds more than 8 surfaces: at most n / 9 of them at any level — sized once, not level by level
        const size_t s_max = (size_t)(n / (kSahLeaf + 1u)) + 1u;
        SAH_TRY(d_segs.reserve(s_max * sizeof(SahSeg)));
        SAH_TRY(d_ce.reserve(s_max * 48));
        SAH_TRY(d_plans.reserve(s_max * sizeof(SahPlan)));
        SAH_TRY(d_count.reserve(s_max * cells * 4));
        SAH_TRY(d_bbox.reserve(s_max * cells * 48));
        SAH_TRY(d_splits.reserve(s_max * sizeof(SahSplit)));
    }

    if (timing) SAH_TRY(hipDeviceSynchronize());
    const auto t1 = now();
    uint32_t levels = 0;
    SahTree T;
    double root_box[6];
    for (int c = 0; c < 3; c++) {
        root_box[c] = s->bb_min[c];
        root_box[3 + c] = s->bb_max[c];
    }
    T.add(root_box, 0u, n <= kSahLeaf ? (uint32_t)n : 0u);
    std::vector<SahSeg> segs, next;
    if (n > kSahLeaf) {
        SahSeg r;
        r.start = 0;
        r.size = (uint32_t)n;
        r.node = 0;
        r.rule = (uint32_t)arity;
        memcpy(r.box, root_box, 48);
        segs.push_back(r);
    }
    std::vector<SahSplit> splits;
    std::vector<Run> runs;
    std::vector<double> arb_box;
    uint32_t* order = d_order.as<uint32_t>();
    uint32_t* order_next = d_order2.as<uint32_t>();
    while (!segs.empty()) {
        const uint32_t S = (uint32_t)segs.size();
        SAH_TRY(d_segs.reserve((size_t)S * sizeof(SahSeg)));
        SAH_TRY(d_ce.reserve((size_t)S * 48));
        SAH_TRY(d_plans.reserve((size_t)S * sizeof(SahPlan)));
        SAH_TRY(d_count.reserve((size_t)S * cells * 4));
        SAH_TRY(d_bbox.reserve((size_t)S * cells * 48));
        SAH_TRY(d_splits.reserve((size_t)S * sizeof(SahSplit)));
        SAH_TRY(hipMemcpy(d_segs.p, segs.data(), (size_t)S * sizeof(SahSeg), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(segmentOfKernel, dim3(gridFor(n)), dim3(256), 0, 0, d_segs.as<SahSeg>(), S, n, d_seg_of.as<uint32_t>());
        hipLaunchKernelGGL(fillKeysKernel, dim3(gridFor((uint64_t)S * 6)), dim3(256), 0, 0, d_ce.as<unsigned long long>(), (uint64_t)S);
        hipLaunchKernelGGL(centroidBoundsKernel, dim3(gridFor(n)), dim3(256), 0, 0, order, d_seg_of.as<uint32_t>(), d_centroid.as<double>(), n,
                           d_ce.as<unsigned long long>());
        hipLaunchKernelGGL(planKernel, dim3(gridFor(S, 64)), dim3(64), 0, 0, d_segs.as<SahSeg>(), S, d_ce.as<unsigned long long>(), d_plans.as<SahPlan>());
        SAH_TRY(hipMemsetAsync(d_count.p, 0, (size_t)S * cells * 4, 0));
        hipLaunchKernelGGL(fillKeysKernel, dim3(gridFor((uint64_t)S * cells * 6)), dim3(256), 0, 0, d_bbox.as<unsigned long long>(), (uint64_t)S * cells);
        hipLaunchKernelGGL(binKernel, dim3(gridFor(n)), dim3(256), 0, 0, order, d_seg_of.as<uint32_t>(), d_centroid.as<double>(), d_bb.as<double>(), n,
                           d_plans.as<SahPlan>(), bins, cells, d_cell_of.as<uint32_t>(), d_count.as<uint32_t>(), d_bbox.as<unsigned long long>());
        hipLaunchKernelGGL(decodeKeysKernel, dim3(gridFor((uint64_t)S * cells * 6)), dim3(256), 0, 0, d_bbox.as<unsigned long long>(), (uint64_t)S * cells * 6);
        hipLaunchKernelGGL(splitKernel, dim3(gridFor(S, 64)), dim3(64), 0, 0, d_segs.as<SahSeg>(), S, d_plans.as<SahPlan>(), bins, cells,
                           d_count.as<uint32_t>(), d_bbox.as<double>(), d_splits.as<SahSplit>());
        hipLaunchKernelGGL(childFlagsKernel, dim3(gridFor(n)), dim3(256), 0, 0, d_seg_of.as<uint32_t>(), d_cell_of.as<uint32_t>(), n, d_splits.as<SahSplit>(),
                           bins, d_child_of.as<uint8_t>(), d_f01.as<unsigned long long>(), d_f23.as<unsigned long long>());
        SAH_TRY(hipGetLastError());
        SAH_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_f01.as<unsigned long long>(), d_s01.as<unsigned long long>(), (int)n));
        SAH_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_f23.as<unsigned long long>(), d_s23.as<unsigned long long>(), (int)n));
        hipLaunchKernelGGL(scatterKernel, dim3(gridFor(n)), dim3(256), 0, 0, order, d_seg_of.as<uint32_t>(), d_child_of.as<uint8_t>(), n, d_segs.as<SahSeg>(),
                           d_splits.as<SahSplit>(), d_s01.as<unsigned long long>(), d_s23.as<unsigned long long>(), order_next);
        SAH_TRY(hipGetLastError());
        std::swap(order, order_next);
        splits.resize(S);
        SAH_TRY(hipMemcpy(splits.data(), d_splits.p, (size_t)S * sizeof(SahSplit), hipMemcpyDeviceToHost));
        // boxes of round-robin parts (rare): the runs in the order sahGrow consumes them
        runs.clear();
        for (uint32_t i = 0; i < S; i++)
            if (splits[i].mode == kSahArb) {
                uint32_t at = segs[i].start;
                for (uint32_t k = 0; k < splits[i].arb; k++) {
                    runs.push_back(Run{at, splits[i].child_size[k]});
                    at += splits[i].child_size[k];
                }
            }
        arb_box.assign(runs.size() * 6, 0.0);
        if (!runs.empty()) {
            SAH_TRY(d_runs.reserve(runs.size() * sizeof(Run)));
            SAH_TRY(d_run_box.reserve(runs.size() * 48));
            SAH_TRY(hipMemcpy(d_runs.p, runs.data(), runs.size() * sizeof(Run), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fillKeysKernel, dim3(gridFor(runs.size() * 6)), dim3(256), 0, 0, d_run_box.as<unsigned long long>(), (uint64_t)runs.size());
            hipLaunchKernelGGL(runBoxKernel, dim3((uint32_t)runs.size()), dim3(256), 0, 0, d_runs.as<Run>(), order, d_bb.as<double>(),
                               d_run_box.as<unsigned long long>());
            hipLaunchKernelGGL(decodeKeysKernel, dim3(gridFor(runs.size() * 6)), dim3(256), 0, 0, d_run_box.as<unsigned long long>(), (uint64_t)runs.size() * 6);
            SAH_TRY(hipGetLastError());
            SAH_TRY(hipMemcpy(arb_box.data(), d_run_box.p, runs.size() * 48, hipMemcpyDeviceToHost));
        }
        sahGrow(T, segs, splits, arb_box.data(), next);
        segs.swap(next);
        levels++;
    }
    const auto t2 = now();
    std::vector<uint32_t> final_order(n);
    SAH_TRY(hipMemcpy(final_order.data(), order, n * 4, hipMemcpyDeviceToHost));
    sahFinish(T, final_order.data(), n, B);
    if (timing)
        fprintf(stderr, "[mcrt sah] %llu surfaces, %u levels, %zu nodes: allocations + upload + boxes %.1f ms, level loop %.1f ms, order back + depth-first numbering %.1f ms\n",
                (unsigned long long)n, levels, T.start.size(), ms(t0, t1), ms(t1, t2), ms(t2, now()));
    return MCRT_OK;
}
