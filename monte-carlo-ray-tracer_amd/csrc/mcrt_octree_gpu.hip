// GPU-assisted photon-map build: the reference's serial insert + compaction
// (photon-mapper.cpp:169-203, octree.cpp:35-80, linear-octree.cpp:202-244) replaced by
//   1. cellCodeKernel   one lane per photon: its root-to-level-21 octant path (mcrt_octree_shared.hpp)
//   2. rocPRIM radix sort of (code, index) pairs (hipcub::DeviceRadixSort, 63 key bits)
//   3. gatherKernel     photons into sorted = depth-first order (32-byte records, one lane per photon)
//   4. host             octants from the sorted codes: a prefix with more than max_node_data photons is an
//                       inner node, its non-empty 3-bit extensions are its children (binary searches only)
//   5. leafBoundsKernel tight box of every leaf (one lane per leaf, <= max_node_data photons each);
//                       inner boxes are merged upwards on the host
// The result is the tree mcrt_photon_map_build() makes — same octants, same boxes, same photons per leaf — with
// the photons of a leaf in code order instead of input order (no query depends on that order).
// HBM-bound integer/byte work: 8 B key + 4 B index per photon through the sort passes, 32 B read + 32 B
// written by the gather, 32 B read by the box pass.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "mcrt_bvh_shared.hpp"
#include "mcrt_internal.hpp"
#include "mcrt_octree_shared.hpp"

using namespace mcrt;

namespace {

struct BoxArgs {
    double mn[3], mx[3];
};

__global__ void cellCodeKernel(const float* photons, uint64_t n, BoxArgs bb, unsigned long long* keys, uint32_t* index) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = photonCellCode(photons + i * 8, bb.mn, bb.mx);
    index[i] = (uint32_t)i;
}

__global__ void gatherKernel(const float4* in, const uint32_t* index, uint64_t n, float4* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t src = index[i];
    out[2 * i] = in[2 * src];
    out[2 * i + 1] = in[2 * src + 1];
}

// BoundingBox::merge(pos) over the photons of one leaf (bounding-box.cpp:66-73); min/max are exact, so the
// order of the photons does not matter.
__global__ void leafBoundsKernel(const float* sorted, const unsigned long long* leaf_start, const uint32_t* leaf_count, uint32_t leaves,
                                 double* out) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= leaves) return;
    double bb[6] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
                    -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
    const float* p = sorted + leaf_start[l] * 8;
    for (uint32_t i = 0; i < leaf_count[l]; i++, p += 8)
        for (int c = 0; c < 3; c++) {
            const double v = (double)p[3 + c];
            if (bb[c] > v) bb[c] = v;
            if (bb[3 + c] < v) bb[3 + c] = v;
        }
    for (int c = 0; c < 6; c++) out[(size_t)l * 6 + c] = bb[c];
}

struct Dev {
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

#define OCT_TRY(call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) { delete M; return ctxFail(ctx, MCRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
    } while (0)

}  // namespace

extern "C" int mcrt_photon_map_build_gpu(mcrt_ctx* ctx, const float* photons, uint64_t num_photons, const double bb_min[3],
                                         const double bb_max[3], uint32_t max_photons_per_leaf, mcrt_photon_map** out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!out || (num_photons && !photons) || !bb_min || !bb_max || max_photons_per_leaf == 0)
        return ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_photon_map_build_gpu: null argument or zero leaf capacity");
    // hipcub::DeviceRadixSort takes the item count as an int
    if (num_photons > 0x7FFFFFFFull) return ctxFail(ctx, MCRT_ERR_UNSUPPORTED, "photon map larger than 2^31-1 photons per GPU (radix sort item count)");
    mcrt_photon_map* M = new mcrt_photon_map();
    if (num_photons == 0) {
        finishMapDesc(M);
        *out = M;
        return MCRT_OK;
    }
    OCT_TRY(hipSetDevice(ctxDevice(ctx)));
    const uint64_t n = num_photons;
    Dev d_in, d_sorted, d_keys, d_keys2, d_idx, d_idx2, d_tmp;
    OCT_TRY(d_in.alloc(n * 32));
    OCT_TRY(d_sorted.alloc(n * 32));
    OCT_TRY(d_keys.alloc(n * 8));
    OCT_TRY(d_keys2.alloc(n * 8));
    OCT_TRY(d_idx.alloc(n * 4));
    OCT_TRY(d_idx2.alloc(n * 4));
    OCT_TRY(hipMemcpy(d_in.p, photons, n * 32, hipMemcpyHostToDevice));
    BoxArgs bb;
    for (int c = 0; c < 3; c++) {
        bb.mn[c] = bb_min[c];
        bb.mx[c] = bb_max[c];
    }
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(cellCodeKernel, dim3(grid), dim3(256), 0, 0, d_in.as<float>(), n, bb, d_keys.as<unsigned long long>(), d_idx.as<uint32_t>());
    OCT_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    OCT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                               d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels));
    OCT_TRY(d_tmp.alloc(tmp_bytes));
    // stable: photons of one level-21 cell stay in input order
    OCT_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                               d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels));
    hipLaunchKernelGGL(gatherKernel, dim3(grid), dim3(256), 0, 0, d_in.as<float4>(), d_idx2.as<uint32_t>(), n, d_sorted.as<float4>());
    OCT_TRY(hipGetLastError());

    std::vector<unsigned long long> keys(n);
    OCT_TRY(hipMemcpy(keys.data(), d_keys2.p, n * 8, hipMemcpyDeviceToHost));
    OctreeAssembler A;
    A.keys = keys.data();
    A.max_node_data = max_photons_per_leaf;
    A.M = M;
    A.node(0, n, 0, true, 0xFFFFFFFFu);
    if (A.too_deep) {  // more than max_node_data photons inside one 2^-21 cell: only the recursive host builder splits that far
        delete M;
        return mcrt_photon_map_build(photons, num_photons, bb_min, bb_max, max_photons_per_leaf, out);
    }
    const uint32_t leaves = (uint32_t)A.leaves.size();
    std::vector<unsigned long long> lstart(leaves);
    std::vector<uint32_t> lcount(leaves);
    for (uint32_t l = 0; l < leaves; l++) {
        lstart[l] = M->start[A.leaves[l]];
        lcount[l] = (uint32_t)M->contained[A.leaves[l]];
    }
    Dev d_ls, d_lc, d_lb;
    OCT_TRY(d_ls.alloc((size_t)leaves * 8));
    OCT_TRY(d_lc.alloc((size_t)leaves * 4));
    OCT_TRY(d_lb.alloc((size_t)leaves * 48));
    OCT_TRY(hipMemcpy(d_ls.p, lstart.data(), (size_t)leaves * 8, hipMemcpyHostToDevice));
    OCT_TRY(hipMemcpy(d_lc.p, lcount.data(), (size_t)leaves * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(leafBoundsKernel, dim3((leaves + 255) / 256), dim3(256), 0, 0, d_sorted.as<float>(), d_ls.as<unsigned long long>(),
                       d_lc.as<uint32_t>(), leaves, d_lb.as<double>());
    OCT_TRY(hipGetLastError());
    std::vector<double> lb((size_t)leaves * 6);
    OCT_TRY(hipMemcpy(lb.data(), d_lb.p, (size_t)leaves * 48, hipMemcpyDeviceToHost));
    M->bounds.assign(M->start.size() * 6, 0.0);
    for (uint32_t l = 0; l < leaves; l++) memcpy(&M->bounds[(size_t)A.leaves[l] * 6], &lb[(size_t)l * 6], 48);
    A.mergeBounds();
    M->photons.resize(n * 8);
    OCT_TRY(hipMemcpy(M->photons.data(), d_sorted.p, n * 32, hipMemcpyDeviceToHost));
    finishMapDesc(M);
    *out = M;
    return MCRT_OK;
}

// ------------------------------------------------------------------------------------------------
// Octree BVH (mcrt_bvh_shared.hpp): per-surface boxes + centroid path codes, radix sort and the gather of the boxes
// on the device; octant assembly, leaf boxes (8 surfaces each at most) and in-leaf ordering on the host.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void surfaceCodeKernel(const uint8_t* kind, const double* surf_v, const double* quadrics, uint64_t n, BoxArgs cube, double* bb,
                                  unsigned long long* keys, uint32_t* index) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double b[6];
    surfaceBounds(kind[i], surf_v + 9 * i, quadrics, b);
    for (int c = 0; c < 6; c++) bb[i * 6 + c] = b[c];
    keys[i] = cellCode((b[3] + b[0]) / 2.0, (b[4] + b[1]) / 2.0, (b[5] + b[2]) / 2.0, cube.mn, cube.mx);
    index[i] = (uint32_t)i;
}

__global__ void gatherBoxKernel(const double* in, const uint32_t* index, uint64_t n, double* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t src = index[i];
    for (int c = 0; c < 6; c++) out[i * 6 + c] = in[src * 6 + c];
}

#define BVH_TRY(call)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return ctxFail(ctx, MCRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

int mcrt::bvhOctreeGpu(mcrt_ctx* ctx, const mcrt_scene_desc* s, mcrt_bvh* B) {
    BVH_TRY(hipSetDevice(ctxDevice(ctx)));
    const uint64_t n = s->num_surfaces;
    if (n > 0x7FFFFFFFull) return ctxFail(ctx, MCRT_ERR_UNSUPPORTED, "more than 2^31-1 surfaces (radix sort item count)");
    Dev d_kind, d_v, d_q, d_bb, d_bb2, d_keys, d_keys2, d_idx, d_idx2, d_tmp;
    BVH_TRY(d_kind.alloc(n));
    BVH_TRY(d_v.alloc(n * 72));
    BVH_TRY(d_q.alloc((size_t)s->num_quadrics * 22 * 8));
    BVH_TRY(d_bb.alloc(n * 48));
    BVH_TRY(d_bb2.alloc(n * 48));
    BVH_TRY(d_keys.alloc(n * 8));
    BVH_TRY(d_keys2.alloc(n * 8));
    BVH_TRY(d_idx.alloc(n * 4));
    BVH_TRY(d_idx2.alloc(n * 4));
    BVH_TRY(hipMemcpy(d_kind.p, s->surf_kind, n, hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_v.p, s->surf_v, n * 72, hipMemcpyHostToDevice));
    if (s->num_quadrics) BVH_TRY(hipMemcpy(d_q.p, s->quadrics, (size_t)s->num_quadrics * 22 * 8, hipMemcpyHostToDevice));
    BoxArgs cube;
    bvhRootCube(s, cube.mn, cube.mx);
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(surfaceCodeKernel, dim3(grid), dim3(256), 0, 0, d_kind.as<uint8_t>(), d_v.as<double>(), d_q.as<double>(), n, cube,
                       d_bb.as<double>(), d_keys.as<unsigned long long>(), d_idx.as<uint32_t>());
    BVH_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    BVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                               d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels));
    BVH_TRY(d_tmp.alloc(tmp_bytes));
    BVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                               d_idx.as<uint32_t>(), d_idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels));
    hipLaunchKernelGGL(gatherBoxKernel, dim3(grid), dim3(256), 0, 0, d_bb.as<double>(), d_idx2.as<uint32_t>(), n, d_bb2.as<double>());
    BVH_TRY(hipGetLastError());
    std::vector<unsigned long long> keys(n);
    std::vector<uint32_t> index(n);
    std::vector<double> sorted_bb(n * 6);
    BVH_TRY(hipMemcpy(keys.data(), d_keys2.p, n * 8, hipMemcpyDeviceToHost));
    BVH_TRY(hipMemcpy(index.data(), d_idx2.p, n * 4, hipMemcpyDeviceToHost));
    BVH_TRY(hipMemcpy(sorted_bb.data(), d_bb2.p, n * 48, hipMemcpyDeviceToHost));
    return assembleOctreeBvh(keys.data(), index.data(), sorted_bb.data(), n, B) ? MCRT_OK : MCRT_ERR_UNSUPPORTED;
}
