// Device-resident photon pass (SURVEY.md §8(f) ranks 1 + 2 without the host round trips): the photon lists the emission
// kernel appends stay in HBM and become the maps the eye pass searches —
//   PhotonMapper::PhotonMapper            integrator/photon-mapper/photon-mapper.cpp:31-203 (emission, Octree inserts)
//   Octree<Photon>::insert                octree/octree.cpp:35-80
//   LinearOctree<Photon>::compact         octree/linear-octree.cpp:202-244
// Steps per map, all on the device (included by mcrt_hip.hip, inside its anonymous namespace):
//   1. cell codes (photonCellCode, mcrt_octree_shared.hpp), radix sort of (code, index), gather of the 32-byte records:
//      the photons in the depth-first order of the compacted octree;
//   2. octants, level by level: a thread per (inner octant of the level, child octant 0..7) finds the child's photon range
//      with two binary searches over the parent's range of sorted codes and appends the non-empty ones (atomic
//      cursor); a child with more than max_node_data photons goes on the next level's list. 21 small launches;
//   3. depth-first numbering: the octants sorted by (first photon, depth) are in pre-order (a parent starts where its
//      first child starts and is shallower); the next sibling of an octant is the first octant starting at its end, if
//      that one has the same depth;
//   4. leaf boxes from the photons, inner boxes merged upwards level by level (min / max: exact, any order);
//   5. the wave search's record lists (WideRec, mcrt_waveknn.hpp): counted per octant, exclusive scan, filled.
// Only counters cross PCIe. The octants, boxes and per-leaf photon sets are those of OctreeAssembler / the reference
// (tests: the device-built map read back and compared with the host builder's).
#pragma once

namespace pdev {

struct BoxArgs {
    double mn[3], mx[3];
};

__global__ void codeKernel(const float* photons, uint64_t n, BoxArgs bb, unsigned long long* keys, uint32_t* index) {
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

// octants in creation (breadth-first) order
struct Nodes {
    uint32_t* start;
    uint32_t* count;
    uint32_t* parent;  // creation index of the parent
    uint8_t* depth;
    uint8_t* leaf;
};

// counters: [0] octants so far, [1] entries of the next level's list, [2] "too deep" flag
__global__ void expandKernel(const unsigned long long* keys, Nodes N, const uint32_t* level_list, uint32_t level_count, int depth,
                             uint32_t max_node_data, uint32_t capacity, uint32_t* next_list, unsigned int* counters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= level_count * 8u) return;
    const uint32_t node = level_list[t >> 3], o = t & 7u;
    const uint32_t lo = N.start[node], hi = lo + N.count[node];
    const int shift = 3 * (kCodeLevels - 1 - depth);
    auto firstAbove = [&](uint32_t digit_below) {  // first index of [lo, hi) whose octant at this level is >= digit_below
        uint32_t a = lo, b = hi;
        while (a < b) {
            const uint32_t mid = a + (b - a) / 2;
            if (((keys[mid] >> shift) & 7ull) < digit_below) a = mid + 1;
            else b = mid;
        }
        return a;
    };
    const uint32_t c0 = firstAbove(o), c1 = o == 7u ? hi : firstAbove(o + 1u);
    if (c1 <= c0) return;  // empty octants are dropped (linear-octree.cpp:225-231)
    const uint32_t id = atomicAdd(&counters[0], 1u);
    if (id >= capacity) return;  // (cannot happen with the capacity the host computes; checked there)
    const uint32_t cnt = c1 - c0;
    const bool is_leaf = cnt <= max_node_data;
    const bool stop = is_leaf || depth + 1 >= kCodeLevels;
    if (!is_leaf && stop) counters[2] = 1u;  // a level-21 cell with more than max_node_data photons: the recursive host builder's case
    N.start[id] = c0;
    N.count[id] = cnt;
    N.parent[id] = node;
    N.depth[id] = (uint8_t)(depth + 1);
    N.leaf[id] = stop ? 1 : 0;
    if (!stop) next_list[atomicAdd(&counters[1], 1u)] = id;
}

__global__ void orderKeyKernel(Nodes N, uint32_t n, unsigned long long* keys, uint32_t* index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((unsigned long long)N.start[i] << 5) | N.depth[i];
    index[i] = i;
}

// creation order -> depth-first order: dfs_of[creation index]
__global__ void inverseKernel(const uint32_t* order, uint32_t n, uint32_t* dfs_of) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dfs_of[order[j]] = j;
}

__global__ void arrangeKernel(Nodes N, const uint32_t* order, const uint32_t* dfs_of, uint32_t n, uint32_t* start, uint32_t* contained,
                              uint8_t* leaf, uint8_t* depth, uint32_t* parent) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t c = order[j];
    start[j] = N.start[c];
    contained[j] = N.count[c];
    leaf[j] = N.leaf[c];
    depth[j] = N.depth[c];
    parent[j] = c == 0u ? 0xFFFFFFFFu : dfs_of[N.parent[c]];
}

__global__ void siblingKernel(const uint32_t* start, const uint32_t* contained, const uint8_t* depth, uint32_t n, uint32_t* next) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t end = start[j] + contained[j];
    uint32_t a = j + 1, b = n;  // first octant after j that starts at or after j's end: the one after j's subtree
    while (a < b) {
        const uint32_t mid = a + (b - a) / 2;
        if (start[mid] < end) a = mid + 1;
        else b = mid;
    }
    next[j] = (a < n && depth[a] == depth[j]) ? a : 0xFFFFFFFFu;
}

// BoundingBox::merge(pos) over the photons of a leaf (bounding-box.cpp:66-73); inner octants start empty
__global__ void leafBoxKernel(const float* sorted, const uint32_t* start, const uint32_t* contained, const uint8_t* leaf, uint32_t n, double* bounds) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double bb[6] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
                    -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
    if (leaf[j]) {
        const float* p = sorted + (size_t)start[j] * 8;
        for (uint32_t i = 0; i < contained[j]; i++, p += 8)
            for (int c = 0; c < 3; c++) {
                const double v = (double)p[3 + c];
                if (bb[c] > v) bb[c] = v;
                if (bb[3 + c] < v) bb[3 + c] = v;
            }
    }
    for (int c = 0; c < 6; c++) bounds[(size_t)j * 6 + c] = bb[c];
}

// BoundingBox::merge(BB) of the children into the inner octants of one depth (bounding-box.cpp:57-64)
__global__ void mergeBoxKernel(const uint8_t* leaf, const uint8_t* depth, const uint32_t* next, uint32_t n, int d, double* bounds) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || leaf[j] || depth[j] != d) return;
    double bb[6];
    for (int c = 0; c < 6; c++) bb[c] = bounds[(size_t)j * 6 + c];
    for (uint32_t ch = j + 1; ch != 0xFFFFFFFFu && ch < n; ch = next[ch])
        for (int c = 0; c < 3; c++) {
            const double lo = bounds[(size_t)ch * 6 + c], hi = bounds[(size_t)ch * 6 + 3 + c];
            if (bb[c] > lo) bb[c] = lo;
            if (bb[3 + c] < hi) bb[3 + c] = hi;
        }
    for (int c = 0; c < 6; c++) bounds[(size_t)j * 6 + c] = bb[c];
}

// ---- record lists of the wave search (what uploadMap builds on the host): an octant that is not scannable (inner, more
// than k photons) lists its scannable children and, for every other child, that child's children
__device__ inline bool scannable(const uint8_t* leaf, const uint32_t* contained, uint32_t o, uint32_t k) { return leaf[o] != 0 || contained[o] <= k; }

__global__ void wideCountKernel(const uint8_t* leaf, const uint32_t* contained, const uint32_t* next, uint32_t n, uint32_t k, uint32_t* count) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    uint32_t cnt = 0;
    if (!scannable(leaf, contained, o, k))
        for (uint32_t c = o + 1; c != 0xFFFFFFFFu && c < n; c = next[c]) {
            if (scannable(leaf, contained, c, k)) cnt++;
            else
                for (uint32_t g = c + 1; g != 0xFFFFFFFFu && g < n; g = next[g]) cnt++;
        }
    count[o] = cnt;
}

__global__ void wideFillKernel(const uint8_t* leaf, const uint32_t* start, const uint32_t* contained, const uint32_t* next, const double* bounds,
                               uint32_t n, uint32_t k, const uint32_t* first, const uint32_t* count, WideRec* wide) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n || scannable(leaf, contained, o, k)) return;
    uint32_t at = first[o];
    auto fill = [&](uint32_t c) {
        WideRec r;
        for (int i = 0; i < 6; i++) r.b[i] = bounds[(size_t)c * 6 + i];
        r.contained = contained[c];
        if (scannable(leaf, contained, c, k)) {
            r.a = start[c];
            r.m = 0x80000000u | contained[c];
        } else {
            r.a = first[c];
            r.m = count[c];
        }
        r.pad = 0u;
        wide[at++] = r;
    };
    for (uint32_t c = o + 1; c != 0xFFFFFFFFu && c < n; c = next[c]) {
        if (scannable(leaf, contained, c, k)) fill(c);
        else
            for (uint32_t g = c + 1; g != 0xFFFFFFFFu && g < n; g = next[g]) fill(g);
    }
}

}  // namespace pdev

// One map from a photon list in device memory (not modified). Installs it as map `which` of the context. timing: optional,
// [0] sort + gather ms, [1] octants ms, [2] boxes + record lists ms (host clock around synchronised sections).
int buildMapOnDevice(mcrt_ctx* ctx, int which, const float* d_photons, uint64_t n, const double bb_min[3], const double bb_max[3],
                     uint32_t max_node_data, double* timing) {
    using namespace pdev;
    PhotonMapView& v = ctx->maps[which];
    memset(&v, 0, sizeof(v));
    ctx->map_children_ptr[which] = nullptr;
    ctx->map_root_a[which] = ctx->map_root_m[which] = 0u;
    if (n == 0) return MCRT_OK;
    if (n > 0x7FFFFFFFull) return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon map larger than 2^31-1 photons per GPU (radix sort item count)");
    // Work buffers come from the context's pool and stay allocated between calls (grow-only): a frame's photon pass used to spend
    // 0.37 of its 0.40 s of map building in hipMalloc / hipFree of multi-GB buffers.
    ctx->pass_pool.rewind();
    hipStream_t st = ctx->stream;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();

    // ---- 1. codes, sort, gather
    DevBuf& keys = ctx->pass_pool.take();
    DevBuf& keys2 = ctx->pass_pool.take();
    DevBuf& idx = ctx->pass_pool.take();
    DevBuf& idx2 = ctx->pass_pool.take();
    DevBuf& tmp = ctx->pass_pool.take();
    HIP_TRY(ctx, keys.reserve(n * 8));
    HIP_TRY(ctx, keys2.reserve(n * 8));
    HIP_TRY(ctx, idx.reserve(n * 4));
    HIP_TRY(ctx, idx2.reserve(n * 4));
    BoxArgs bb;
    for (int c = 0; c < 3; c++) {
        bb.mn[c] = bb_min[c];
        bb.mx[c] = bb_max[c];
    }
    const uint32_t grid_n = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(codeKernel, dim3(grid_n), dim3(256), 0, st, d_photons, n, bb, keys.as<unsigned long long>(), idx.as<uint32_t>());
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), idx.as<uint32_t>(),
                                                    idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels, st));
    HIP_TRY(ctx, tmp.reserve(tmp_bytes));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), idx.as<uint32_t>(),
                                                    idx2.as<uint32_t>(), (int)n, 0, 3 * kCodeLevels, st));  // stable
    HIP_TRY(ctx, ctx->map_photons[which].reserve(n * 32));
    hipLaunchKernelGGL(gatherKernel, dim3(grid_n), dim3(256), 0, st, reinterpret_cast<const float4*>(d_photons), idx2.as<uint32_t>(), n,
                       ctx->map_photons[which].as<float4>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    auto t1 = now();

    // ---- 2. octants, level by level (an inner octant holds > max_node_data photons; chains of single children are at most 21 long)
    const uint64_t cap64 = n + 21ull * (n / ((uint64_t)max_node_data + 1ull)) + 64ull;
    if (cap64 > 0xFFFFFFF0ull) return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon map too large for 32-bit octant indices");
    const uint32_t cap = (uint32_t)cap64;
    DevBuf& n_start = ctx->pass_pool.take();
    DevBuf& n_count = ctx->pass_pool.take();
    DevBuf& n_parent = ctx->pass_pool.take();
    DevBuf& n_depth = ctx->pass_pool.take();
    DevBuf& n_leaf = ctx->pass_pool.take();
    DevBuf& list_a = ctx->pass_pool.take();
    DevBuf& list_b = ctx->pass_pool.take();
    DevBuf& counters = ctx->pass_pool.take();
    HIP_TRY(ctx, n_start.reserve((size_t)cap * 4));
    HIP_TRY(ctx, n_count.reserve((size_t)cap * 4));
    HIP_TRY(ctx, n_parent.reserve((size_t)cap * 4));
    HIP_TRY(ctx, n_depth.reserve(cap));
    HIP_TRY(ctx, n_leaf.reserve(cap));
    const size_t list_cap = (size_t)(n / ((uint64_t)max_node_data + 1ull)) + 64;  // inner octants of one level: disjoint, > max_node_data photons each
    HIP_TRY(ctx, list_a.reserve(list_cap * 4));
    HIP_TRY(ctx, list_b.reserve(list_cap * 4));
    HIP_TRY(ctx, counters.reserve(4 * sizeof(unsigned int)));
    Nodes N{n_start.as<uint32_t>(), n_count.as<uint32_t>(), n_parent.as<uint32_t>(), n_depth.as<uint8_t>(), n_leaf.as<uint8_t>()};
    {   // root
        const bool root_leaf = n <= max_node_data;
        const uint32_t r_start = 0u, r_count = (uint32_t)n, r_parent = 0u, zero = 0u;
        const uint8_t r_depth = 0, r_leaf = root_leaf ? 1 : 0;
        const unsigned int c0[4] = {1u, 0u, 0u, 0u};
        HIP_TRY(ctx, hipMemcpy(n_start.p, &r_start, 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(n_count.p, &r_count, 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(n_parent.p, &r_parent, 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(n_depth.p, &r_depth, 1, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(n_leaf.p, &r_leaf, 1, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(counters.p, c0, sizeof(c0), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(list_a.p, &zero, 4, hipMemcpyHostToDevice));
        uint32_t level_count = root_leaf ? 0u : 1u;
        DevBuf* cur = &list_a;
        DevBuf* nxt = &list_b;
        for (int depth = 0; depth < kCodeLevels && level_count > 0; depth++) {
            hipLaunchKernelGGL(expandKernel, dim3((level_count * 8u + 255u) / 256u), dim3(256), 0, st, keys2.as<unsigned long long>(), N, cur->as<uint32_t>(),
                               level_count, depth, max_node_data, cap, nxt->as<uint32_t>(), counters.as<unsigned int>());
            HIP_TRY(ctx, hipGetLastError());
            unsigned int h[4];
            HIP_TRY(ctx, hipMemcpyAsync(h, counters.p, sizeof(h), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (h[0] > cap) return fail(ctx, MCRT_ERR_HIP, "photon octree: octant capacity exceeded");
            if (h[2]) {  // (buildMapAnyDepth, mcrt_hip.hip, takes it from here: the recursive host builder has no level limit)
                ctx->dense_cell_refused = true;
                return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon octree: more than max_photons_per_leaf photons in one 2^-21 cell of the map's cube");
            }
            level_count = h[1];
            if (level_count > list_cap) return fail(ctx, MCRT_ERR_HIP, "photon octree: level list capacity exceeded");
            HIP_TRY(ctx, hipMemsetAsync(counters.as<unsigned int>() + 1, 0, 4, st));  // (no host source that could leave scope before the copy runs)
            std::swap(cur, nxt);
        }
    }
    unsigned int hc[4];
    HIP_TRY(ctx, hipMemcpyAsync(hc, counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));  // on the context's stream, behind its kernels
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint32_t no = hc[0];  // octants

    // ---- 3. depth-first order
    DevBuf& okeys = ctx->pass_pool.take();
    DevBuf& okeys2 = ctx->pass_pool.take();
    DevBuf& order0 = ctx->pass_pool.take();
    DevBuf& order = ctx->pass_pool.take();
    DevBuf& dfs_of = ctx->pass_pool.take();
    DevBuf& depth = ctx->pass_pool.take();
    DevBuf& parent = ctx->pass_pool.take();
    HIP_TRY(ctx, okeys.reserve((size_t)no * 8));
    HIP_TRY(ctx, okeys2.reserve((size_t)no * 8));
    HIP_TRY(ctx, order0.reserve((size_t)no * 4));
    HIP_TRY(ctx, order.reserve((size_t)no * 4));
    HIP_TRY(ctx, dfs_of.reserve((size_t)no * 4));
    HIP_TRY(ctx, depth.reserve(no));
    HIP_TRY(ctx, parent.reserve((size_t)no * 4));
    const uint32_t grid_o = (no + 255u) / 256u;
    hipLaunchKernelGGL(orderKeyKernel, dim3(grid_o), dim3(256), 0, st, N, no, okeys.as<unsigned long long>(), order0.as<uint32_t>());
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, okeys.as<unsigned long long>(), okeys2.as<unsigned long long>(), order0.as<uint32_t>(),
                                                    order.as<uint32_t>(), (int)no, 0, 37, st));
    HIP_TRY(ctx, tmp.reserve(tmp_bytes));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, okeys.as<unsigned long long>(), okeys2.as<unsigned long long>(), order0.as<uint32_t>(),
                                                    order.as<uint32_t>(), (int)no, 0, 37, st));
    hipLaunchKernelGGL(inverseKernel, dim3(grid_o), dim3(256), 0, st, order.as<uint32_t>(), no, dfs_of.as<uint32_t>());
    HIP_TRY(ctx, ctx->map_start[which].reserve((size_t)no * 4));
    HIP_TRY(ctx, ctx->map_contained[which].reserve((size_t)no * 4));
    HIP_TRY(ctx, ctx->map_next[which].reserve((size_t)no * 4));
    HIP_TRY(ctx, ctx->map_leaf[which].reserve(no));
    HIP_TRY(ctx, ctx->map_bounds[which].reserve((size_t)no * 48));
    uint32_t* m_start = ctx->map_start[which].as<uint32_t>();
    uint32_t* m_cont = ctx->map_contained[which].as<uint32_t>();
    uint32_t* m_next = ctx->map_next[which].as<uint32_t>();
    uint8_t* m_leaf = ctx->map_leaf[which].as<uint8_t>();
    double* m_bounds = ctx->map_bounds[which].as<double>();
    hipLaunchKernelGGL(arrangeKernel, dim3(grid_o), dim3(256), 0, st, N, order.as<uint32_t>(), dfs_of.as<uint32_t>(), no, m_start, m_cont, m_leaf,
                       depth.as<uint8_t>(), parent.as<uint32_t>());
    hipLaunchKernelGGL(siblingKernel, dim3(grid_o), dim3(256), 0, st, m_start, m_cont, depth.as<uint8_t>(), no, m_next);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    auto t2 = now();

    // ---- 4. boxes
    hipLaunchKernelGGL(leafBoxKernel, dim3(grid_o), dim3(256), 0, st, ctx->map_photons[which].as<float>(), m_start, m_cont, m_leaf, no, m_bounds);
    for (int d = kCodeLevels - 1; d >= 0; d--)
        hipLaunchKernelGGL(mergeBoxKernel, dim3(grid_o), dim3(256), 0, st, m_leaf, depth.as<uint8_t>(), m_next, no, d, m_bounds);
    HIP_TRY(ctx, hipGetLastError());

    // ---- 5. record lists of the wave search
    {
        const uint32_t k = std::max<uint32_t>(ctx->k_nearest, 1u);
        DevBuf& cnt = ctx->pass_pool.take();
        DevBuf& first = ctx->pass_pool.take();
        HIP_TRY(ctx, cnt.reserve((size_t)no * 4));
        HIP_TRY(ctx, first.reserve((size_t)no * 4));
        hipLaunchKernelGGL(wideCountKernel, dim3(grid_o), dim3(256), 0, st, m_leaf, m_cont, m_next, no, k, cnt.as<uint32_t>());
        HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt.as<uint32_t>(), first.as<uint32_t>(), (int)no, st));
        HIP_TRY(ctx, tmp.reserve(tmp_bytes));
        HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, cnt.as<uint32_t>(), first.as<uint32_t>(), (int)no, st));
        uint32_t last_first = 0, last_cnt = 0, root_cnt = 0, root_first = 0, root_cont = 0;
        uint8_t root_leaf = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&last_first, first.as<uint32_t>() + (no - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&last_cnt, cnt.as<uint32_t>() + (no - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&root_cnt, cnt.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&root_first, first.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&root_cont, m_cont, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&root_leaf, m_leaf, 1, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        const uint64_t total = (uint64_t)last_first + last_cnt;
        if (total > 0xFFFFFFFFull) return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon map too large for 32-bit record indices");
        HIP_TRY(ctx, ctx->map_children[which].reserve((size_t)std::max<uint64_t>(total, 1) * sizeof(WideRec)));
        hipLaunchKernelGGL(wideFillKernel, dim3(grid_o), dim3(256), 0, st, m_leaf, m_start, m_cont, m_next, m_bounds, no, k, first.as<uint32_t>(),
                           cnt.as<uint32_t>(), ctx->map_children[which].as<WideRec>());
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));
        ctx->map_children_ptr[which] = ctx->map_children[which].as<WideRec>();
        const bool root_scan = root_leaf != 0 || root_cont <= k;
        ctx->map_root_a[which] = root_scan ? 0u : root_first;
        ctx->map_root_m[which] = root_scan ? (0x80000000u | root_cont) : root_cnt;
    }
    auto t3 = now();
    v.num_octants = no;
    v.num_photons = n;
    v.octant_bounds = m_bounds;
    v.octant_start = m_start;
    v.octant_contained = m_cont;
    v.octant_next = m_next;
    v.octant_leaf = m_leaf;
    v.photons = ctx->map_photons[which].as<float>();
    if (int rc = buildMapPositions(ctx, which, n)) return rc;
    if (timing) {
        timing[0] += ms(t0, t1);
        timing[1] += ms(t1, t2);
        timing[2] += ms(t2, t3);
    }
    return MCRT_OK;
}
