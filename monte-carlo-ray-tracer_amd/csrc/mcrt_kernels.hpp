// gfx950 kernels of libmcrt_hip.so and what they share (kernel parameter blocks, LDS plans, staging, wave helpers).
// Included by mcrt_hip.hip only, inside its anonymous namespace: the host side (context, launches, C ABI) lives there.
//
// Kernel structure of the megakernels (one launch per render):
//   persistent workgroups; every LANE owns one pixel at a time and runs that pixel's spp samples in
//   the reference's order (so the per-pixel FP64 sum is the reference's, film.cpp:99-113). The loop
//   body is ONE BOUNCE (mcrt_integrator.hpp); a lane whose path ended is re-armed with the next
//   sample in the same iteration, and a lane whose pixel is complete takes the next pixel from a
//   global work counter through a wave-aggregated pop: __ballot of the lanes that need work, one
//   atomicAdd of popcount by the first such lane, prefix-popcount as each lane's offset. The wave
//   therefore stays compacted (all 64 lanes tracing) until the frame runs out of pixels.
//   Pixels are enumerated in 8x8 tiles so that a wave's rays are coherent.
//   LDS per workgroup: Sobol byte tables (24 KiB), per-lane traversal stack [depth][lane], and a
//   staged copy of the scene: the whole scene when it is small (hexagon_room: 16 nodes, 44 prims),
//   else the top of the BVH (breadth-first prefix of the node array / of the quantised child blocks).
// The wavefront pipeline (trace / shade / kNN launches over a slot pool in HBM) is described in mcrt_wavefront.hpp.
#pragma once

// ------------------------------------------------------------------------------------------------
// kernel parameter blocks
// ------------------------------------------------------------------------------------------------
struct DeviceScene {
    // global memory
    uint32_t num_nodes, num_surfaces, num_materials, num_lights;
    const double* node_bounds;
    const NodeMeta* node_meta;
    const Node64* nodes64;
    const QBlock* qblocks;        // quantised child blocks of the trace kernel (mcrt_qbvh.hpp)
    uint32_t num_qblocks, q_root_a, q_root_m;
    uint32_t stack_depth;         // traversal-stack entries per lane (LDS + spill slab): max(kMaxStackDepth, the tree's stack bound), mcrt_upload_scene
    uint32_t q_nodes;             // records in nodes64 = num_nodes, or — scene without a BVH — the nodes of the index-range tree the wavefront pipeline walks (mcrt_layout.hpp)
    const double* prim;
    const double* flat_prim;      // kind-sorted copy (flat mode)
    const uint32_t* flat_index;
    uint32_t flat_tris;
    const float* flat_pre;        // FP32 cull records of the flat loop (mcrt_scene.hpp), null = none
    uint32_t pre_tri_pairs, pre_sph_pairs;
    double pre_centre[3], pre_bound;
    const double* surf_v;
    const double* surf_normal;
    const double* surf_rec;  // [n][16] one 128-byte shading record per surface: normal, material | kind << 32, vertex normals (ShadeViewT::surf_rec)
    const double* surf_vn;  // may be null
    const double* surf_area;
    const uint32_t* surf_material;
    const uint8_t* surf_kind;
    const mcrt_material* materials;
    const uint32_t* light_surface;
    const double* light_cdf;
    const uint32_t* sobol_tab;
    double scene_ior;
    // staging plan
    uint32_t stage_all;    // 1: whole scene in LDS
    uint32_t stage_nodes;  // number of leading nodes staged
    uint32_t flat;         // 1: tiny scene, test every primitive in a wave-uniform loop (no BVH walk)
};

struct DevicePhotonMap {
    PhotonMapView view;
};

struct RenderParams {
    mcrt_camera_desc cam;
    uint32_t global_seed, spp;
    uint32_t owned_rows;
    uint32_t tiles_x, tiles_y;
    uint64_t work_items;  // tiles_x * tiles_y * 64
    unsigned long long* work_counter;
    unsigned long long* stats;  // paths, rays, node_tests, prim_tests, knn_searches, overflow, knn_octants
    StackEntry* spill;
    uint32_t total_lanes;
    uint32_t knn_max_visit;  // per-lane photon search: frontier entries per lane (KnnScratch::max_visit; in the padding before the maps)
    // photon mapping
    PhotonMapView global_map, caustic_map;
    uint32_t k_nearest, direct_visualization;
    double* knn_res_d2;
    uint32_t* knn_res_idx;
    double* knn_visit_d2;
    uint32_t* knn_visit_oct;
    // lane-state-machine gating (renderKernelSM)
    int sm_shade_lanes, sm_regen_lanes, sm_min_trav, sm_leaf_lanes, sm_min_inner, sm_lds_depth;
    // Sample-chunked work units (renderKernel, renderKernelSM). A unit is one pixel's samples [c*chunk, (c+1)*chunk): with
    // whole pixels as units a frame ends with most lanes idle while the last pixels run their spp samples (1080p over 8
    // GPUs leaves 2 pixels per resident lane: 70 % efficiency), with chunks the tail is one chunk. Every sample's radiance is
    // stored and sampleResolveKernel adds a pixel's samples in the reference's order, so the FP64 sum is still
    // Film::deposit's. A launch covers the local rows [row_base, row_end) — one pass of the frame, sized to the store.
    uint32_t chunk, chunk_shift;  // samples per unit; units per pixel = 1 << chunk_shift
    uint32_t row_base, row_end;
    uint64_t pass_pixels;         // (row_end - row_base) * width
    double* samples;              // [spp][pass_pixels][3]
};

struct WorkUnit {
    uint32_t lx, ly, first, end;  // pixel (column, local row; edge tiles hold positions outside the image), samples [first, end)
};
__device__ inline WorkUnit decodeUnit(const RenderParams& prm, unsigned long long w) {
    const unsigned long long item = w >> prm.chunk_shift;  // pixels in 8x8 tiles
    const uint32_t c = (uint32_t)(w - (item << prm.chunk_shift));
    const uint32_t tile = (uint32_t)(item >> 6), in = (uint32_t)(item & 63u);
    WorkUnit u;
    u.lx = (tile % prm.tiles_x) * 8u + (in & 7u);
    u.ly = prm.row_base + (tile / prm.tiles_x) * 8u + (in >> 3);
    u.first = c * prm.chunk;
    u.end = u.first + prm.chunk < prm.spp ? u.first + prm.chunk : prm.spp;
    return u;
}
// Film::deposit with the default box filter: own pixel, weight 1 (film.cpp:13-17,61-79,99-105) — the addend, kept per sample
__device__ inline void storeSample(const RenderParams& prm, uint32_t sample, uint32_t px, uint32_t ly, d3 radiance) {
    double* o = prm.samples + ((size_t)sample * prm.pass_pixels + ((size_t)(ly - prm.row_base) * prm.cam.width + px)) * 3;
    o[0] = radiance.x * 1.0;
    o[1] = radiance.y * 1.0;
    o[2] = radiance.z * 1.0;
}

// ------------------------------------------------------------------------------------------------
// LDS carving
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kStatsWords = 8 + 2 * kNumPhases;
// stats[5] counts two things the host tells apart: lanes whose traversal stack overflowed (one each; cannot happen - the stacks are sized
// to the tree's own bound) and searches whose kNN frontier did (kKnnOverflowFlag and above: octrees with leaves far smaller than k)
constexpr unsigned long long kKnnOverflowUnit = 1ull << 32;
constexpr uint32_t kBlock = 512;  // 8 waves per workgroup, one workgroup per CU (LDS-bound, see planLds)

constexpr uint32_t kPmLdsIors = 2;  // refraction-history entries per lane the 1024-lane photon-mapping kernel keeps in LDS

struct LdsPlan {
    uint32_t sobol, stack, iors, node_bounds, node_meta, prim, flat_prim, flat_index, flat_pre, surf_v, surf_normal, surf_vn, surf_area, surf_material,
        surf_kind, materials, light_surface, light_cdf, total;
};

__host__ __device__ inline uint32_t alignUp(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

__host__ __device__ inline LdsPlan planLds(const DeviceScene& s, uint32_t block, bool with_stack = true, uint32_t stack_depth = kLdsStackDepth,
                                           uint32_t iors_depth = kMaxIors) {
    LdsPlan p;
    uint32_t off = 0;
    p.sobol = off; off += kSobolTableWords * 4;
    p.stack = off; off += (with_stack && !s.flat) ? stack_depth * block * (uint32_t)sizeof(StackEntry) : 0u;  // the flat loop has no stack
    p.iors = off; off += iors_depth * block * 8u;
    off = alignUp(off, 16);
    const uint32_t nn = s.stage_all ? s.num_nodes : s.stage_nodes;
    p.node_bounds = off; off += nn * 48;
    p.node_meta = off; off = alignUp(off + nn * 8, 16);
    if (s.stage_all) {
        const uint32_t ns = s.num_surfaces;
        p.prim = off; off += s.flat ? 0 : ns * kPrimStride * 8;  // flat scenes only read the kind-sorted copy
        p.flat_prim = off; off += s.flat ? ns * kPrimStride * 8 : 0;
        p.flat_index = off; off = alignUp(off + (s.flat ? ns * 4 : 0), 16);
        p.flat_pre = off; off += (s.flat && s.flat_pre) ? (s.pre_tri_pairs * (uint32_t)kTriPairFloats + s.pre_sph_pairs * (uint32_t)kSphPairFloats) * 4u : 0u;
        p.surf_v = off; off += ns * 72;
        p.surf_normal = off; off += ns * 24;
        p.surf_vn = off; off += (s.surf_vn ? ns * 72 : 0);
        p.surf_area = off; off += ns * 8;
        p.surf_material = off; off = alignUp(off + ns * 4, 16);
        p.surf_kind = off; off = alignUp(off + ns, 16);
        p.materials = off; off = alignUp(off + s.num_materials * (uint32_t)sizeof(mcrt_material), 16);
        p.light_cdf = off; off += s.num_lights * 8;
        p.light_surface = off; off = alignUp(off + s.num_lights * 4, 16);
    } else {
        p.prim = p.flat_prim = p.flat_index = p.flat_pre = p.surf_v = p.surf_normal = p.surf_vn = p.surf_area = p.surf_material = p.surf_kind = p.materials =
            p.light_cdf = p.light_surface = off;
    }
    p.total = off;
    return p;
}

template <class T>
__device__ inline MCRT_LDS_AS T* ldsAt(unsigned char* base, uint32_t off) {
    return (MCRT_LDS_AS T*)((MCRT_LDS_AS unsigned char*)base + off);  // offset added in address space 3
}

template <class T>
__device__ inline void stageCopy(MCRT_LDS_AS T* dst, const T* src, uint32_t count) {
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
}

// Builds the per-lane views; stages the scene into LDS (ends with __syncthreads()).
// kAll: whole scene LDS-resident (the views carry address-space-3 pointers, so every scene access in
// the hot loops is a ds_read); otherwise only the top of the BVH is staged.
// kPreK: the flat loop reads its cull records from the kernel's argument block (renderKernelFlatK): the LDS copy is not filled (its
// 2.8 KB stay carved - planLds is shared with the host's launch geometry - but no workgroup spends a staging loop on them).
template <bool kAll, bool kFlat = false, bool kPreK = false>
__device__ inline void setupViews(const DeviceScene& s, unsigned char* lds, SceneViewT<kAll>& sv, ShadeViewT<kAll>& sh,
                                  SobolTab& tab, LaneStack& stk, RefractionHistory& rh, StackEntry* spill, uint32_t total_lanes,
                                  uint32_t stack_depth = kLdsStackDepth, double* iors_global = nullptr) {
    const LdsPlan p = planLds(s, blockDim.x, !kFlat, stack_depth, iors_global ? kPmLdsIors : (uint32_t)kMaxIors);
    rh.giors = iors_global;
    rh.glane = blockIdx.x * blockDim.x + threadIdx.x;
    rh.gstride = total_lanes;
    rh.lds_depth = iors_global ? (int)kPmLdsIors : kMaxIors;
    rh.iors = ldsAt<double>(lds, p.iors) + threadIdx.x;
    rh.stride = blockDim.x;
    rh.size = 0;
    MCRT_LDS_AS uint32_t* ltab = ldsAt<uint32_t>(lds, p.sobol);
    stageCopy(ltab, s.sobol_tab, (uint32_t)kSobolTableWords);
    if constexpr (!kAll) glibc235::stageSinCosTab();  // (kernels of LDS-resident scenes read the sin/cos table from memory: mcrt_libm.hpp)
    tab = ltab;

    stk.lds = ldsAt<StackEntry>(lds, p.stack) + threadIdx.x;
    stk.lds_stride = blockDim.x;
    stk.spill = spill + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    stk.spill_stride = total_lanes;
    stk.max_depth = (int)s.stack_depth;

    sv.num_nodes = s.flat ? 0u : s.num_nodes;
    sv.num_surfaces = s.num_surfaces;
    const uint32_t nn = kAll ? s.num_nodes : s.stage_nodes;
    MCRT_LDS_AS double* lnb = ldsAt<double>(lds, p.node_bounds);
    MCRT_LDS_AS NodeMeta* lnm = ldsAt<NodeMeta>(lds, p.node_meta);
    stageCopy(lnb, s.node_bounds, nn * 6);
    stageCopy(lnm, s.node_meta, nn);
    sv.lds_nodes = nn;
    sv.lds_node_bounds = lnb;
    sv.lds_node_meta = lnm;
    sh.num_lights = s.num_lights;
    sh.scene_ior = s.scene_ior;

    if constexpr (kAll) {
        const uint32_t ns = s.num_surfaces;
        sv.node_bounds = lnb;
        sv.node_meta = lnm;
        MCRT_LDS_AS double* lp = ldsAt<double>(lds, p.prim);
        if (!s.flat) stageCopy(lp, s.prim, ns * kPrimStride);
        sv.prim = lp;
        sv.flat_tris = s.flat_tris;
        sv.flat_prim = nullptr;
        sv.flat_index = nullptr;
        // (always a valid LDS address — the pair counts gate its use: hipcc 7.2 fails to select code for a read through a
        // null address-space-3 pointer plus an offset)
        MCRT_LDS_AS float* lpre = ldsAt<float>(lds, p.flat_pre);
        const bool cull = s.flat && s.flat_pre;
        sv.flat_pre = lpre;
        sv.pre_tri_pairs = cull ? s.pre_tri_pairs : 0u;
        sv.pre_sph_pairs = cull ? s.pre_sph_pairs : 0u;
        sv.pre_cx = s.pre_centre[0];
        sv.pre_cy = s.pre_centre[1];
        sv.pre_cz = s.pre_centre[2];
        sv.pre_bound = s.pre_bound;
        if (cull && !kPreK) stageCopy(lpre, s.flat_pre, s.pre_tri_pairs * (uint32_t)kTriPairFloats + s.pre_sph_pairs * (uint32_t)kSphPairFloats);
        if (s.flat) {
            MCRT_LDS_AS double* lfp = ldsAt<double>(lds, p.flat_prim);
            stageCopy(lfp, s.flat_prim, ns * kPrimStride);
            sv.flat_prim = lfp;
            MCRT_LDS_AS uint32_t* lfi = ldsAt<uint32_t>(lds, p.flat_index);
            stageCopy(lfi, s.flat_index, ns);
            sv.flat_index = lfi;
        }
        MCRT_LDS_AS double* lv = ldsAt<double>(lds, p.surf_v);
        stageCopy(lv, s.surf_v, ns * 9);
        sh.surf_v = lv;
        MCRT_LDS_AS double* ln = ldsAt<double>(lds, p.surf_normal);
        stageCopy(ln, s.surf_normal, ns * 3);
        sh.surf_normal = ln;
        sh.surf_rec = nullptr;
        MCRT_LDS_AS double* lvn = ldsAt<double>(lds, p.surf_vn);
        if (s.surf_vn) stageCopy(lvn, s.surf_vn, ns * 9);
        sh.surf_vn = lvn;
        MCRT_LDS_AS double* la = ldsAt<double>(lds, p.surf_area);
        stageCopy(la, s.surf_area, ns);
        sh.surf_area = la;
        MCRT_LDS_AS uint32_t* lm = ldsAt<uint32_t>(lds, p.surf_material);
        stageCopy(lm, s.surf_material, ns);
        sh.surf_material = lm;
        MCRT_LDS_AS uint8_t* lk = ldsAt<uint8_t>(lds, p.surf_kind);
        stageCopy(lk, s.surf_kind, ns);
        sh.surf_kind = lk;
        MCRT_LDS_AS uint64_t* lmat = ldsAt<uint64_t>(lds, p.materials);
        stageCopy(lmat, reinterpret_cast<const uint64_t*>(s.materials), s.num_materials * (uint32_t)(sizeof(mcrt_material) / 8));
        sh.materials = (MCRT_LDS_AS const mcrt_material*)lmat;
        MCRT_LDS_AS double* lc = ldsAt<double>(lds, p.light_cdf);
        stageCopy(lc, s.light_cdf, s.num_lights);
        sh.light_cdf = lc;
        MCRT_LDS_AS uint32_t* ll = ldsAt<uint32_t>(lds, p.light_surface);
        stageCopy(ll, s.light_surface, s.num_lights);
        sh.light_surface = ll;
    } else {
        sv.node_bounds = s.node_bounds;
        sv.node_meta = s.node_meta;
        sv.prim = s.prim;
        sv.flat_tris = 0;
        sv.flat_prim = nullptr;
        sv.flat_index = nullptr;
        sv.flat_pre = nullptr;
        sv.pre_tri_pairs = sv.pre_sph_pairs = 0;
        sv.pre_cx = sv.pre_cy = sv.pre_cz = sv.pre_bound = 0.0;
        sh.surf_v = s.surf_v;
        sh.surf_normal = s.surf_normal;
        sh.surf_rec = s.surf_rec;
        sh.surf_vn = s.surf_vn;
        sh.surf_area = s.surf_area;
        sh.surf_material = s.surf_material;
        sh.surf_kind = s.surf_kind;
        sh.materials = s.materials;
        sh.light_surface = s.light_surface;
        sh.light_cdf = s.light_cdf;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// wave helpers (wavefront = 64 lanes on gfx950)
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t laneId() { return __lane_id(); }

__device__ inline unsigned long long waveBroadcast64(unsigned long long v, int src) {  // src wave-uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// Wave-aggregated pop from a global counter: lanes with need == true each receive a distinct index.
__device__ inline unsigned long long wavePop(bool need, unsigned long long* counter) {
    const unsigned long long mask = waveBallot(need);
    if (mask == 0ull) return 0ull;
    const int leader = __ffsll((long long)mask) - 1;
    const uint32_t lane = laneId();
    unsigned long long base = 0ull;
    if ((int)lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
    base = waveBroadcast64(base, leader);
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    return base + rank;
}

__device__ inline void waveAccumulate(unsigned long long* dst, uint32_t v) {
    unsigned long long x = v;
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo = __shfl_down((uint32_t)x, off, 64), hi = __shfl_down((uint32_t)(x >> 32), off, 64);
        x += ((unsigned long long)hi << 32) | lo;
    }
    if (laneId() == 0 && x) atomicAdd(dst, x);
}

// ------------------------------------------------------------------------------------------------
// the integrator kernel
// ------------------------------------------------------------------------------------------------
// kFlat != 0: instance for flat-mode scenes only (path tracer): no BVH walk in the code, no traversal stack in LDS;
// kFlat == 2: 768-lane workgroups (3 waves per SIMD, 168 VGPRs); kFlat == 3: 1024 lanes (4 waves per SIMD, 128 VGPRs).
// pre_k (kPreK): the flat loop's cull records as part of the kernel's argument block (renderKernelFlatK below), or null.
template <int kIntegrator, bool kCount, bool kAll, bool kProf, int kFlat, bool kPreK>
__device__ __forceinline__ void renderKernelBody(const DeviceScene& scene, const RenderParams& prm, const float* pre_k) {
    MCRT_DYNAMIC_LDS(lds, 16);
    SceneViewT<kAll> sv;
    ShadeViewT<kAll> sh;
    SobolTab tab;
    LaneStack stk;
    RefractionHistory rh;
    setupViews<kAll, kFlat != 0, kPreK>(scene, lds, sv, sh, tab, stk, rh, prm.spill, prm.total_lanes);
    sv.flat_pre_k = pre_k;

    const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
    KnnScratch ks;
    ks.overflowed = 0u;
    ks.max_visit = 0u;
    PhotonViews pv;
    if (kIntegrator == MCRT_INTEGRATOR_PHOTON_MAPPER) {
        ks.res_d2 = prm.knn_res_d2 + gl;
        ks.res_idx = prm.knn_res_idx + gl;
        ks.visit_d2 = prm.knn_visit_d2 + gl;
        ks.visit_oct = prm.knn_visit_oct + gl;
        ks.stride = prm.total_lanes;
        ks.max_visit = prm.knn_max_visit;
        pv.global_map = prm.global_map;
        pv.caustic_map = prm.caustic_map;
        pv.k_nearest = prm.k_nearest;
        pv.direct_visualization = prm.direct_visualization != 0;
    }

    PathState st;
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    PhaseProf<kProf> prof;
    if constexpr (kProf) prof.begin();
    uint32_t paths = 0, searches = 0, octant_visits = 0;
    bool have_pixel = false, path_active = false, exhausted = false;
    uint32_t px = 0, py = 0, ly = 0, sample = 0, sample_end = 0;
    const uint32_t W = prm.cam.width;

    for (;;) {
        if (kProf) prof.mark(kPhLoop);
        const bool need = !have_pixel && !exhausted;
        if (waveBallot(need)) {
            const unsigned long long w = wavePop(need, prm.work_counter);
            if (need) {
                if (w >= prm.work_items) {
                    exhausted = true;
                } else {
                    const WorkUnit u = decodeUnit(prm, w);
                    if (u.lx < W && u.ly < prm.row_end && u.first < u.end) {
                        px = u.lx;
                        ly = u.ly;
                        py = localToGlobalRow(prm.cam, ly);
                        have_pixel = true;
                        sample = u.first;
                        sample_end = u.end;
                        st.smp.initiate(prm.global_seed, py * W + px);  // camera.cpp:73
                    }
                }
            }
        }
        if (!waveBallot(have_pixel)) {
            if (!waveBallot(!exhausted)) break;
            continue;
        }
        if (have_pixel) {
            if (!path_active) {
                if (kProf) prof.mark(kPhRegen);
                st.smp.setIndex(sample);  // camera.cpp:77
                pathBegin(st, rh, cameraRay<!kAll>(prm.cam, sh.scene_ior, px, py, st.smp, tab));
                path_active = true;
                paths++;
            }
            bool done;
            if (kIntegrator == MCRT_INTEGRATOR_PHOTON_MAPPER)
                done = photonMapperBounce<kCount, kAll>(st, rh, sv, sh, pv, stk, ks, cnt, searches, octant_visits, tab);
            else
                done = pathTracerBounce<kCount, kAll, kProf, kFlat != 0, kPreK>(st, rh, sv, sh, stk, cnt, tab, &prof);
            if (kProf) prof.mark(kPhLoop);
            if (done) {
                storeSample(prm, sample, px, ly, st.radiance);
                path_active = false;
                if (++sample == sample_end) have_pixel = false;
            }
        }
    }

    waveAccumulate(prm.stats + 0, paths);
    waveAccumulate(prm.stats + 1, cnt.rays);
    if (kCount) {
        waveAccumulate(prm.stats + 2, cnt.node_tests);
        waveAccumulate(prm.stats + 3, cnt.prim_tests);
    }
    waveAccumulate(prm.stats + 4, searches);
    if constexpr (kIntegrator == MCRT_INTEGRATOR_PHOTON_MAPPER) waveAccumulate(prm.stats + 5, cnt.overflow | ks.overflowed);  // (a search that ran out of frontier)
    else waveAccumulate(prm.stats + 5, cnt.overflow);
    waveAccumulate(prm.stats + 7, rh.overflow ? 1u : 0u);
    waveAccumulate(prm.stats + 6, octant_visits);
    if constexpr (kProf) {
        prof.mark(kPhLoop);
        for (int i = 0; i < kNumPhases; i++) {
            atomicAdd(prm.stats + 8 + i, prof.wave_cycles[i]);
            atomicAdd(prm.stats + 8 + kNumPhases + i, prof.lane_cycles[i]);
        }
    }
}
template <int kIntegrator, bool kCount, bool kAll, bool kProf = false, int kFlat = 0>
__global__ void __launch_bounds__(kFlat == 3 ? 1024 : kFlat == 2 ? 768 : kBlock) renderKernel(const DeviceScene scene, const RenderParams prm) {
    renderKernelBody<kIntegrator, kCount, kAll, kProf, kFlat, false>(scene, prm, nullptr);
}
// The flat megakernel with its cull records in the ARGUMENT BLOCK (round 5; scenes whose records fit kFlatPreArgFloats: C1, C2,
// C2-GGX): sceneIntersect's kPreK. 512 lanes like kFlat == 1. An argument block holds 4 KB; DeviceScene + RenderParams take under 1 KB.
constexpr uint32_t kFlatPreArgFloats = 704;  // 2 816 bytes: e.g. 14 triangle pairs + 16 sphere pairs, or 22 triangle pairs
struct FlatPreArg {
    float v[kFlatPreArgFloats];
};
static_assert(sizeof(DeviceScene) + sizeof(RenderParams) + sizeof(FlatPreArg) + 64 <= 4096, "the argument block of renderKernelFlatK");
template <int kLanes = (int)kBlock>  // 512, 768 (3 waves per SIMD, 168 VGPRs) or 1024 (4 waves, 128 VGPRs): MCRT_FLAT_BLOCK
__global__ void __launch_bounds__(kLanes) renderKernelFlatK(const DeviceScene scene, const RenderParams prm, const FlatPreArg pre) {
    renderKernelBody<MCRT_INTEGRATOR_PATH_TRACER, false, true, false, kLanes == 1024 ? 3 : kLanes == 768 ? 2 : 1, true>(scene, prm, pre.v);
}

// ------------------------------------------------------------------------------------------------
// the lane-state-machine integrator (scenes whose BVH is walked; see mcrt_lanesm.hpp)
// ------------------------------------------------------------------------------------------------
struct SmLdsPlan {
    uint32_t sobol, stack, iors, nodes, prim, surf_v, surf_normal, surf_vn, surf_area, surf_material, surf_kind, materials,
        light_surface, light_cdf, total;
};

__host__ __device__ inline SmLdsPlan planSmLds(const DeviceScene& s, uint32_t block, uint32_t stack_depth = kLdsStackDepth) {
    SmLdsPlan p;
    uint32_t off = 0;
    p.sobol = off; off += kSobolTableWords * 4;
    p.stack = off; off += stack_depth * block * (uint32_t)sizeof(SmStackEntry);
    p.iors = off; off += kMaxIors * block * 8u;
    off = alignUp(off, 64);
    const uint32_t nn = s.stage_all ? s.num_nodes : s.stage_nodes;
    p.nodes = off; off += nn * 64u;
    if (s.stage_all) {
        const uint32_t ns = s.num_surfaces;
        p.prim = off; off += ns * kPrimStride * 8;
        p.surf_v = off; off += ns * 72;
        p.surf_normal = off; off += ns * 24;
        p.surf_vn = off; off += (s.surf_vn ? ns * 72 : 0);
        p.surf_area = off; off += ns * 8;
        p.surf_material = off; off = alignUp(off + ns * 4, 16);
        p.surf_kind = off; off = alignUp(off + ns, 16);
        p.materials = off; off = alignUp(off + s.num_materials * (uint32_t)sizeof(mcrt_material), 16);
        p.light_cdf = off; off += s.num_lights * 8;
        p.light_surface = off; off = alignUp(off + s.num_lights * 4, 16);
    } else {
        p.prim = p.surf_v = p.surf_normal = p.surf_vn = p.surf_area = p.surf_material = p.surf_kind = p.materials =
            p.light_cdf = p.light_surface = off;
    }
    p.total = off;
    return p;
}

enum : int { kStRegen = 0, kStTrav = 1, kStShade = 2, kStDone = 3 };

// kLanes: workgroup size (512 / 768 / 1024 = 2 / 3 / 4 waves per SIMD at 256 / 168 / 128 VGPRs); the larger workgroups keep
// fewer stack entries per lane in LDS (prm.sm_lds_depth).
template <bool kCount, bool kAll, bool kProf = false, int kLanes = (int)kBlock>
__global__ void __launch_bounds__(kLanes) renderKernelSM(const DeviceScene scene, const RenderParams prm) {
    MCRT_DYNAMIC_LDS(lds, 16);
    const SmLdsPlan p = planSmLds(scene, blockDim.x, (uint32_t)prm.sm_lds_depth);

    // ---- staging
    MCRT_LDS_AS uint32_t* ltab = ldsAt<uint32_t>(lds, p.sobol);
    stageCopy(ltab, scene.sobol_tab, (uint32_t)kSobolTableWords);
    if constexpr (!kAll) glibc235::stageSinCosTab();
    const SobolTab tab = ltab;
    SmStack stk;
    stk.lds = ldsAt<SmStackEntry>(lds, p.stack) + threadIdx.x;
    stk.lds_stride = blockDim.x;
    stk.spill = reinterpret_cast<SmStackEntry*>(prm.spill) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    stk.spill_stride = prm.total_lanes;
    stk.lds_depth = prm.sm_lds_depth;
    stk.max_depth = (int)scene.stack_depth;
    RefractionHistory rh;
    rh.iors = ldsAt<double>(lds, p.iors) + threadIdx.x;
    rh.stride = blockDim.x;
    rh.size = 0;

    SmSceneView<kAll> sv;
    ShadeViewT<kAll> sh;
    sv.num_nodes = scene.num_nodes;
    const uint32_t nn = kAll ? scene.num_nodes : scene.stage_nodes;
    MCRT_LDS_AS uint64_t* lnodes = ldsAt<uint64_t>(lds, p.nodes);
    // whole scene resident: the exact 64-byte records; tree in HBM: quantised child blocks (mcrt_qbvh.hpp), the
    // first nn of them (the top of the tree) here in LDS
    QView<true> qv;
    qv.blocks = scene.qblocks;
    qv.lds_blocks = 0;
    qv.lds_ptr = (MCRT_LDS_AS const QBlock*)lnodes;
    qv.root_a = scene.q_root_a;
    qv.root_m = scene.q_root_m;
    if constexpr (kAll) {
        stageCopy(lnodes, reinterpret_cast<const uint64_t*>(scene.nodes64), nn * 8u);
        sv.lds_nodes = nn;
    } else {
        qv.lds_blocks = nn < scene.num_qblocks ? nn : scene.num_qblocks;
        stageCopy(lnodes, reinterpret_cast<const uint64_t*>(scene.qblocks), qv.lds_blocks * 8u);
        sv.lds_nodes = 0;
    }
    sv.lds_node_ptr = (MCRT_LDS_AS const Node64*)lnodes;
    sh.num_lights = scene.num_lights;
    sh.scene_ior = scene.scene_ior;
    if constexpr (kAll) {
        const uint32_t ns = scene.num_surfaces;
        sv.nodes = (MCRT_LDS_AS const Node64*)lnodes;
        MCRT_LDS_AS double* lp = ldsAt<double>(lds, p.prim);
        stageCopy(lp, scene.prim, ns * kPrimStride);
        sv.prim = lp;
        MCRT_LDS_AS double* lv = ldsAt<double>(lds, p.surf_v);
        stageCopy(lv, scene.surf_v, ns * 9);
        sh.surf_v = lv;
        MCRT_LDS_AS double* ln = ldsAt<double>(lds, p.surf_normal);
        stageCopy(ln, scene.surf_normal, ns * 3);
        sh.surf_normal = ln;
        sh.surf_rec = nullptr;
        MCRT_LDS_AS double* lvn = ldsAt<double>(lds, p.surf_vn);
        if (scene.surf_vn) stageCopy(lvn, scene.surf_vn, ns * 9);
        sh.surf_vn = lvn;
        MCRT_LDS_AS double* la = ldsAt<double>(lds, p.surf_area);
        stageCopy(la, scene.surf_area, ns);
        sh.surf_area = la;
        MCRT_LDS_AS uint32_t* lm = ldsAt<uint32_t>(lds, p.surf_material);
        stageCopy(lm, scene.surf_material, ns);
        sh.surf_material = lm;
        MCRT_LDS_AS uint8_t* lk = ldsAt<uint8_t>(lds, p.surf_kind);
        stageCopy(lk, scene.surf_kind, ns);
        sh.surf_kind = lk;
        MCRT_LDS_AS uint64_t* lmat = ldsAt<uint64_t>(lds, p.materials);
        stageCopy(lmat, reinterpret_cast<const uint64_t*>(scene.materials), scene.num_materials * (uint32_t)(sizeof(mcrt_material) / 8));
        sh.materials = (MCRT_LDS_AS const mcrt_material*)lmat;
        MCRT_LDS_AS double* lc = ldsAt<double>(lds, p.light_cdf);
        stageCopy(lc, scene.light_cdf, scene.num_lights);
        sh.light_cdf = lc;
        MCRT_LDS_AS uint32_t* ll = ldsAt<uint32_t>(lds, p.light_surface);
        stageCopy(ll, scene.light_surface, scene.num_lights);
        sh.light_surface = ll;
    } else {
        sv.nodes = scene.nodes64;
        sv.prim = scene.prim;
        sh.surf_v = scene.surf_v;
        sh.surf_normal = scene.surf_normal;
        sh.surf_rec = scene.surf_rec;
        sh.surf_vn = scene.surf_vn;
        sh.surf_area = scene.surf_area;
        sh.surf_material = scene.surf_material;
        sh.surf_kind = scene.surf_kind;
        sh.materials = scene.materials;
        sh.light_surface = scene.light_surface;
        sh.light_cdf = scene.light_cdf;
    }
    __syncthreads();

    // ---- per-lane state
    PathState st;
    NeePending nee;
    nee.pending = false;
    Trav T;
    T.active = false;
    T.shadow = false;
    T.sp = 0;
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    uint32_t paths = 0;
    int state = kStRegen;
    bool have_pixel = false, alive = false;
    uint32_t px = 0, py = 0, ly = 0, sample = 0, sample_end = 0;
    const uint32_t W = prm.cam.width;

    // thresholds: the expensive blocks run when this many lanes wait for them, or when fewer than
    // kMinTrav lanes are still traversing (so that nothing starves)
    const int kShadeLanes = prm.sm_shade_lanes, kRegenLanes = prm.sm_regen_lanes, kMinTrav = prm.sm_min_trav;
    const int kLeafLanes = prm.sm_leaf_lanes, kMinInner = prm.sm_min_inner;
    PhaseProf<kProf> prof;  // phases here: regen, traverse = inner steps, shade, shadow = leaf steps, loop = transitions
    if constexpr (kProf) prof.begin();

    auto smBegin = [&](d3 o, d3 d, d3 inv, bool shadow, const ShadowQuery* sq, TraceCounters& c) {
        if constexpr (kAll) travBegin<kAll, kCount>(sv, T, o, d, inv, shadow, sq, c);
        else travBeginQ<kAll, true, kCount>(sv, qv, T, o, d, inv, shadow, sq, c);
    };
    // path end: Film::deposit (box filter) + next sample / work unit bookkeeping
    auto endPath = [&]() {
        storeSample(prm, sample, px, ly, st.radiance);
        if (++sample == sample_end) have_pixel = false;
        state = kStRegen;
    };

    for (;;) {
        unsigned long long tp = prof.now();
        // ---- cheap transitions of lanes whose traversal has just finished
        if (state == kStTrav && !T.active) {
            if (T.shadow) {
                smNeeFinish(st, sh, nee, T.best);
                nee.pending = false;
                if (alive) smBegin( st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr, cnt);
                else endPath();
            } else {
                state = kStShade;
            }
        }

        if (kProf) { prof.span(kPhLoop, tp, true); tp = prof.now(); }
        const unsigned long long m_trav = waveBallot(state == kStTrav && T.active);
        const int n_trav = __popcll(m_trav);
        const unsigned long long m_shade = waveBallot(state == kStShade);
        const unsigned long long m_regen = waveBallot(state == kStRegen);
        if (!(m_trav | m_shade | m_regen)) break;  // every lane is done

        // ---- regenerate: next sample of the lane's pixel, or a new pixel from the global counter
        if (m_regen && (__popcll(m_regen) >= kRegenLanes || n_trav < kMinTrav)) {
            const bool need = state == kStRegen && !have_pixel;
            if (waveBallot(need)) {
                const unsigned long long w = wavePop(need, prm.work_counter);
                if (need) {
                    if (w >= prm.work_items) {
                        state = kStDone;
                    } else {
                        const WorkUnit u = decodeUnit(prm, w);
                        if (u.lx < W && u.ly < prm.row_end && u.first < u.end) {
                            px = u.lx;
                            ly = u.ly;
                            py = localToGlobalRow(prm.cam, ly);
                            have_pixel = true;
                            sample = u.first;
                            sample_end = u.end;
                            st.smp.initiate(prm.global_seed, py * W + px);
                        }
                    }
                }
            }
            if (state == kStRegen && have_pixel) {
                st.smp.setIndex(sample);
                pathBegin(st, rh, cameraRay<!kAll>(prm.cam, sh.scene_ior, px, py, st.smp, tab));
                paths++;
                st.smp.shuffle();  // path-tracer.cpp:23, first bounce
                smBegin( st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr, cnt);
                state = kStTrav;
            }
        }

        if (kProf) { prof.span(kPhRegen, tp, (m_regen >> __lane_id()) & 1ull); tp = prof.now(); }
        // ---- shade
        if (m_shade && (__popcll(m_shade) >= kShadeLanes || n_trav < kMinTrav)) {
            if (state == kStShade) {
                Ray shadow_ray;
                ShadowQuery shadow_q;
                alive = smShade(st, rh, sh, T.best, nee, shadow_ray, shadow_q, tab);
                if (alive) st.smp.shuffle();  // top of the next while(true) iteration (path-tracer.cpp:23)
                if (nee.pending) {
                    smBegin( shadow_ray.start, shadow_ray.direction, shadow_ray.inv_direction, true, &shadow_q, cnt);
                    state = kStTrav;
                } else if (alive) {
                    smBegin( st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr, cnt);
                    state = kStTrav;
                } else {
                    endPath();
                }
            }
        }

        if (kProf) { prof.span(kPhShade, tp, (m_shade >> __lane_id()) & 1ull); tp = prof.now(); }
        // ---- traversal steps: inner nodes first (all such lanes together), then leaves
        {
            const bool trav = state == kStTrav && T.active;
            const bool inner = trav && (T.node_m & kSmInner);
            if constexpr (kAll) {
                if (inner) travInnerStep<kAll, kCount>(sv, T, stk, cnt);
            } else {
                if (inner && T.fast) travInnerStepQ<true, kCount>(qv, T, stk, cnt);
                if (inner && !T.fast) travInnerStep<kAll, kCount>(sv, T, stk, cnt);  // zero direction component: exact records
            }
            if (kProf) { prof.span(kPhTraverse, tp, inner); tp = prof.now(); }
            const bool leaf = state == kStTrav && T.active && !(T.node_m & kSmInner);
            const unsigned long long m_leaf = waveBallot(leaf);
            const unsigned long long m_inner = waveBallot(state == kStTrav && T.active && (T.node_m & kSmInner));
            if (m_leaf && (__popcll(m_leaf) >= kLeafLanes || __popcll(m_inner) < kMinInner)) {
                if (leaf) travLeafStep<kAll, kCount>(sv, T, stk, cnt);
            }
            if (kProf) prof.span(kPhShadow, tp, leaf);
        }
    }

    waveAccumulate(prm.stats + 0, paths);
    waveAccumulate(prm.stats + 1, cnt.rays);
    if (kCount) {
        waveAccumulate(prm.stats + 2, cnt.node_tests);
        waveAccumulate(prm.stats + 3, cnt.prim_tests);
    }
    waveAccumulate(prm.stats + 5, cnt.overflow);
    waveAccumulate(prm.stats + 7, rh.overflow ? 1u : 0u);
    if constexpr (kProf) {
        for (int i = 0; i < kNumPhases; i++) {
            atomicAdd(prm.stats + 8 + i, prof.wave_cycles[i]);
            atomicAdd(prm.stats + 8 + kNumPhases + i, prof.lane_cycles[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wavefront path tracer (mcrt_wavefront.hpp): trace kernel + shade kernel, path state pooled in HBM
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kWfBlock = 256;      // shade kernel
constexpr uint32_t kTraceMaxBlock = 1024;  // trace kernel: up to 16 waves per workgroup, 128 VGPRs

struct WfTraceArgs {
    const unsigned long long* count;    // number of queued rays
    unsigned long long* pop;            // next queue index to hand out (zeroed by the shade launch)
    unsigned long long* stats;
    const Node64* nodes;                // exact records: root test, rays with a zero direction component
    const QBlock* qblocks;
    uint32_t num_nodes, lds_blocks, q_root_a, q_root_m;
    const double* prim;
    SmStackEntry* spill;
    uint32_t total_lanes;
    int refill_lanes, leaf_lanes, min_inner, lds_stack;
    int leaf_items;                     // shared leaf step: offered primitives from which a step is issued (MCRT_WF_LEAF_ITEMS)
    uint32_t max_stack;                 // stack entries per lane in all (DeviceScene::stack_depth)
    uint32_t deal_shift;                // queue entries are dealt to the workgroups in blocks of 2^deal_shift
};

// Where the rays of a trace launch come from and where their hits go.
struct PoolRays {  // the wavefront pipeline: rays in queue order (WfRayQueue), hits into the slot pool
    WfPool pool;
    WfRayQueue q;
    __device__ uint32_t load(unsigned long long w, d3& o, d3& d, bool& shadow, ShadowQuery& sq) const { return q.get(w, o, d, shadow, sq); }
    __device__ void store(uint32_t it, const Hit& h) const { wfStoreHit(pool, it, h); }
};
struct ArrayRays {  // mcrt_intersect: closest hits of n rays given as arrays
    const double* start;
    const double* direction;
    double* out_t;
    uint32_t* out_surface;
    double* out_uv;
    __device__ uint32_t load(unsigned long long w, d3& o, d3& d, bool& shadow, ShadowQuery& sq) const {
        o = ld3(start + 3 * (size_t)w);
        d = ld3(direction + 3 * (size_t)w);
        shadow = false;
        sq.t_near = 0.0;
        sq.t_far = kDblMax;
        sq.light = kNoSurface;
        return (uint32_t)w;
    }
    __device__ void store(uint32_t it, const Hit& h) const {
        out_t[it] = h.t;
        out_surface[it] = h.surface;
        out_uv[2 * (size_t)it] = h.u;
        out_uv[2 * (size_t)it + 1] = h.v;
    }
};

#include "mcrt_sharedleaf.hpp"  // the shared leaf step and the wave-synchronous walks (also run on the host: tests/emu/wave_emu.hpp)

// Persistent waves; every lane owns one ray at a time and takes the next one from the queue as soon as its
// traversal has finished (refills are batched: refill_lanes idle lanes, or nothing left to do). Inner nodes
// are visited through quantised child blocks, the top of the tree from LDS.
// Leaves are DEFERRED (mcrt_lanesm.hpp: a lane parks the leaf it reaches and keeps walking) and tested by the WHOLE WAVE
// (travSharedLeafStep, mcrt_sharedleaf.hpp; the default since round 4). The forms this replaced - a lane waiting at its leaf (round 2),
// a pending leaf tested by its own lane (round 3), eight-wide nodes, a slot-scheduled workgroup, two half pools on two streams - lost
// every A/B they were in and were removed in round 6 (measurements: profiles/NOTES_r01_r03.md ... NOTES_r05.md).
// kLean (round 5): bit 0 = the inner visit is travInnerStepQLean (mcrt_qbvh.hpp: FP32 ray kept in the Trav, the three pushes as one
// block of unconditional LDS writes), bit 1 = ... and the tree has no node with more than four children (one block per visit).
// kLean == 0 (round 4's visit) serves MCRT_COUNT_TESTS and MCRT_WF_LEAN=0.
template <class Rays, bool kCount, int kLean = 0>
__global__ void __launch_bounds__(kTraceMaxBlock) wfTraceKernel(const WfTraceArgs a, const Rays rays) {
    MCRT_DYNAMIC_LDS(lds, 64);
    MCRT_LDS_AS QBlock* lq = ldsAt<QBlock>(lds, 0);
    for (uint32_t i = threadIdx.x; i < a.lds_blocks * 16u; i += blockDim.x)
        reinterpret_cast<MCRT_LDS_AS uint32_t*>(lq)[i] = reinterpret_cast<const uint32_t*>(a.qblocks)[i];
    SmStack stk;
    stk.lds = ldsAt<SmStackEntry>(lds, a.lds_blocks * 64u) + threadIdx.x;
    stk.lds_stride = blockDim.x;
    stk.spill = a.spill + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    stk.spill_stride = a.total_lanes;
    stk.lds_depth = a.lds_stack;
    stk.max_depth = (int)a.max_stack;
    SmSceneView<false> sv;
    sv.num_nodes = a.num_nodes;
    sv.nodes = a.nodes;
    sv.prim = a.prim;
    sv.lds_nodes = 1;            // the root (below)
    sv.lds_node_ptr = nullptr;   // set once the root is staged
    QView<true> qv;
    qv.blocks = a.qblocks;
    qv.lds_blocks = a.lds_blocks;
    qv.lds_ptr = lq;
    qv.root_a = a.q_root_a;
    qv.root_m = a.q_root_m;
    // The queue is dealt to the workgroups in blocks of 2^deal_shift consecutive entries, round robin (workgroup g takes
    // blocks g, g + G, g + 2G, ...), and a workgroup's waves take entries from its blocks through a cursor in LDS. (One
    // global cursor for all waves was the kernel's bottleneck: a refill every 32 rays is ~45 M atomics per second on one
    // address, about what an L2 channel serves — refilling at 16 idle lanes instead of 32 cost 37 % of the frame. One
    // contiguous share per workgroup has no atomics either, but neighbouring entries are rays from the same part of the
    // image and the parts differ in cost: the same rays traced 12 % faster shuffled than in pixel order and 20 % slower
    // sorted by origin; dealt in blocks of 64 every order gains and pixel order is the fastest (round 2's ray-sort probe, profiles/r02_ray_sort_probe.log).
    // Interleaved blocks keep a refill's reads consecutive and give every workgroup a sample of the whole queue. Numbering a round's
    // blocks XCD by XCD - the workgroups that share an L2 taking 2048+ NEIGHBOURING rays - changes nothing: +-0.1 % on C3 / C4 /
    // spaceship, round 5, profiles/r05_ab_trace_deal_xcd.log.)
    const unsigned long long n = *a.count;
    MCRT_LDS_AS uint32_t* cursor = ldsAt<uint32_t>(lds, a.lds_blocks * 64u + (uint32_t)a.lds_stack * blockDim.x * (uint32_t)sizeof(SmStackEntry));
    // (behind the cursor's 64 bytes: the rank -> lane map of every wave's shared leaf steps)
    MCRT_LDS_AS uint8_t* share_map = reinterpret_cast<MCRT_LDS_AS uint8_t*>(cursor) + 64u + (threadIdx.x >> 6) * kShareMapBytes;
    // (... and behind the maps the root's exact record: every refill tests it - an LDS read instead of a memory round trip per refill)
    MCRT_LDS_AS Node64* lds_root = reinterpret_cast<MCRT_LDS_AS Node64*>(reinterpret_cast<MCRT_LDS_AS uint8_t*>(cursor) + 64u + (blockDim.x >> 6) * kShareMapBytes);
    if (threadIdx.x < 16u) reinterpret_cast<MCRT_LDS_AS uint32_t*>(lds_root)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.nodes)[threadIdx.x];
    const uint32_t deal_shift = a.deal_shift, deal_mask = (1u << deal_shift) - 1u;
    if (threadIdx.x == 0) *cursor = 0u;
    sv.lds_node_ptr = lds_root;
    __syncthreads();
    auto dealt = [&](uint32_t v) -> unsigned long long {  // the v-th entry of this workgroup
        return ((((unsigned long long)(v >> deal_shift) * gridDim.x + blockIdx.x) << deal_shift) | (v & deal_mask));
    };

    Trav T;
    T.active = false;
    T.shadow = false;
    T.fast = true;
    T.sp = 0;
    LeanRay LR;  // kLean: the ray in FP32 and floatAbove(best.t), kept per ray (travInnerStepQLean)
    LR.of[0] = LR.of[1] = LR.of[2] = LR.invf[0] = LR.invf[1] = LR.invf[2] = LR.best_up = 0.0f;
    PendLeaf P;  // deferred leaves (mcrt_lanesm.hpp): a lane parks the leaf it reaches and keeps walking
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    // MCRT_COUNT_TESTS: where the wave's issue slots go (per wave: iterations, lanes holding a ray, inner / leaf steps and the lanes
    // that took part, leaf lanes kept waiting by the gate, wave cycles inside the two steps and in all)
    unsigned long long ph_iter = 0, ph_have = 0, ph_in_steps = 0, ph_in_lanes = 0, ph_lf_steps = 0, ph_lf_lanes = 0, ph_lf_wait = 0, ph_in_cyc = 0, ph_lf_cyc = 0;
    unsigned long long ph_refill_cyc = 0, ph_pop_cyc = 0;  // (what "rest" is made of: the refill blocks, the loop's pop site)
    const unsigned long long ph_begin = kCount ? clock64() : 0ull;
    bool have = false, exhausted = dealt(0u) >= n;
    uint32_t item = 0;
    for (;;) {
        // Finished rays (deferred leaves: none pending, nothing left to pop) hand their hits back in BATCHES - together with the refill
        // that replaces them (a store per iteration for the three rays that finish in it cost the whole wave ~30 instructions each time)
        const bool done = have && !T.active && P.n == 0u && !T.need_pop;
        const unsigned long long m_done = waveBallot(done);
        unsigned long long m_have = waveBallot(have) & ~m_done;
        const bool want_refill = !exhausted && (64 - __popcll(m_have) >= a.refill_lanes || m_have == 0ull);
        if (m_done && (want_refill || exhausted)) {
            if (done) {
                rays.store(item, T.best);
                have = false;
            }
        } else {
            m_have |= m_done;  // (they keep their lanes until the next batch)
        }
        if (want_refill) {
            const unsigned long long t_refill = kCount ? clock64() : 0ull;
            unsigned long long w = n;
            {
                const unsigned long long need = waveBallot(!have);  // (not empty here)
                const int leader = __ffsll((long long)need) - 1;
                uint32_t base = 0u;
                if ((int)laneId() == leader) base = __atomic_fetch_add(cursor, (uint32_t)__popcll(need), __ATOMIC_RELAXED);  // ds_add_rtn_u32
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                // (the cursor cannot wrap: a workgroup's entries number < 2^32 / G + a block, and it stops within 2^20 of the end)
                if (!have) w = dealt(base + (uint32_t)__popcll(need & ((1ull << laneId()) - 1ull)));
            }
            if (!have && w < n) {
                d3 o, d;
                bool shadow;
                ShadowQuery sq;
                item = rays.load(w, o, d, shadow, sq);
                travBeginQ<false, true, kCount>(sv, qv, T, o, d, rcp3(d), shadow, &sq, cnt);
                if constexpr (kLean != 0) leanRayBegin(LR, T);
                have = true;
            }
            exhausted = waveBallot(!have) != 0ull;  // a lane came back empty-handed: the queue is drained
            if (kCount) ph_refill_cyc += clock64() - t_refill;
        }
        if (!waveBallot(have)) {
            if (exhausted) break;
            continue;
        }
        // A lane standing at a leaf whose pending slot is free parks the leaf; it and every lane whose last visit kept no child take
        // their next node from the stack HERE - the loop's one pop site (round 4; the walk used to pop in three places per
        // iteration - after each of two parking sites and at the end of the inner step - each a chain of dependent LDS reads the
        // whole wave waits for). A lane that pops a leaf while its slot is taken waits; with a free slot it parks it next time round.
        if (have && T.active && !(T.node_m & kSmInner) && P.n == 0u) {
            P.a = T.node_a;
            P.n = T.node_m;
            T.active = false;
            T.need_pop = true;
        }
        if (waveBallot(have && T.need_pop)) {
            const unsigned long long t_pop = kCount ? clock64() : 0ull;
            if (have && T.need_pop) {
                travPopCached(T, stk);
                T.need_pop = false;
            }
            if (kCount) ph_pop_cyc += clock64() - t_pop;
        }
        const bool inner = have && T.active && (T.node_m & kSmInner);
        unsigned long long ti = 0ull;
        if (kCount) {
            const unsigned long long mi = waveBallot(inner);
            ph_iter++;
            ph_have += __popcll(waveBallot(have));
            ph_in_steps += mi ? 1u : 0u;
            ph_in_lanes += __popcll(mi);
            ti = clock64();
        }
        if constexpr ((kLean & 1) != 0) {
            if (inner && T.fast) travInnerStepQLean<true, kCount, (kLean & 2) != 0>(qv, T, LR, stk, cnt);
        } else {
            if (inner && T.fast) travInnerStepQ<true, kCount, true>(qv, T, stk, cnt);
        }
        if (inner && !T.fast) travInnerStep<false, kCount, true>(sv, T, stk, cnt);  // zero direction component: exact records
        if (kCount) ph_in_cyc += clock64() - ti;
        const bool pend = have && P.n != 0u;
        const unsigned long long m_pend = waveBallot(pend);
        const unsigned long long m_inner = waveBallot(have && (T.need_pop || (T.active && (T.node_m & kSmInner))));
        ShareOffer so;
        bool go = m_pend && (__popcll(m_pend) >= a.leaf_lanes || __popcll(m_inner) < a.min_inner);
        // the shared step is gated by what it would TEST: enough offered primitives to fill the wave
        // (counting the offers only for a step that is going to run - the item gate is off by default - was tried in round 5: the
        // extra branch cost the kernel a spilled register and 20 instructions)
        so = shareOffer(pend, P);
        go = go || (int)so.total >= a.leaf_items;
        if (go) {
            unsigned long long tc = 0ull;
            if (kCount) {
                ph_lf_steps++;
                ph_lf_lanes += __popcll(m_pend);
                tc = clock64();
            }
            travSharedLeafStep<kCount>(sv, T, P, so, share_map, cnt);
            if constexpr (kLean != 0) LR.best_up = floatAbove(T.best.t);  // (the step may have improved the hit)
            if (kCount) ph_lf_cyc += clock64() - tc;
        } else if (kCount) {
            ph_lf_wait += __popcll(m_pend);
        }
    }
    if (kCount && laneId() == 0) {
        atomicAdd(a.stats + 8, ph_iter);
        atomicAdd(a.stats + 9, ph_have);
        atomicAdd(a.stats + 10, ph_in_steps);
        atomicAdd(a.stats + 11, ph_in_lanes);
        atomicAdd(a.stats + 12, ph_lf_steps);
        atomicAdd(a.stats + 13, ph_lf_lanes);
        atomicAdd(a.stats + 14, ph_lf_wait);
        atomicAdd(a.stats + 15, ph_in_cyc);
        atomicAdd(a.stats + 16, ph_lf_cyc);
        atomicAdd(a.stats + 17, (unsigned long long)(clock64() - ph_begin));
        atomicAdd(a.stats + 18, ph_refill_cyc);
        atomicAdd(a.stats + 19, ph_pop_cyc);
    }
    waveAccumulate(a.stats + 1, cnt.rays);
    if (kCount) {
        waveAccumulate(a.stats + 2, cnt.node_tests);
        waveAccumulate(a.stats + 3, cnt.prim_tests);
    }
    waveAccumulate(a.stats + 5, cnt.overflow);
}

struct WfShadeArgs {
    WfPool pool;
    uint32_t slot_base, slot_count;   // the slots this launch serves (one half of the pool per stream)
    WfFrame fr;
    WfRayQueue queue;
    unsigned long long* count_out;    // rays queued by this launch
    unsigned long long* count_reset;  // the other parity's counter, consumed by the trace launch before this one
    unsigned long long* pop_reset;
    unsigned long long* work;         // the frame's pixel work counter
    unsigned long long* stats;
    // photon mapper: estimate requests (slot | need_global << 31) for the next kNN launch, and the results of the last one
    uint32_t* requests;
    unsigned long long* rcount_out;
    unsigned long long* rcount_reset;
    unsigned long long* rpop_reset;
    WfPmView pm;
    double* stage;                    // photon mapper: staged Interactions for the kNN launch (DevWfEnv::stage), or null
    uint32_t lds_tables;              // bytes of LDS after the refraction histories for the materials and the light tables (0: read from memory)
};

// The small tables every bounce reads — materials, light distribution — as LDS copies of the shade launch's workgroups: the
// bounce code reads a material record in six to ten places, each a dependent round trip when the record is in memory (L2 hits,
// but the launch is bound by the length of exactly that chain). The views keep generic pointers; the accesses become flat loads.
__host__ __device__ inline uint32_t wfShadeTableBytes(uint32_t num_materials, uint32_t num_lights) {
    return alignUp(num_materials * (uint32_t)sizeof(mcrt_material), 16) + alignUp(num_lights * 8u, 16) + alignUp(num_lights * 4u, 16);
}
constexpr uint32_t kWfShadeTableMax = 12u * 1024u;

struct DevWfEnv {
    unsigned long long* work;
    WfRayQueue queue;
    unsigned long long* count;
    uint32_t* requests;
    unsigned long long* rcount;
    double* stage_buf;                   // [slots][kStageDoubles] staged Interactions of the estimate requests (null: searches only)
    const mcrt_material* mat_view;       // the shade launch's materials (its LDS copy) and the array in memory, for the staged pointer
    const mcrt_material* mat_global;
    __device__ void stage_(uint32_t slot, const InteractionT<false>& ia) const {
        InteractionT<false> q = ia;
        q.material = mat_global + (ia.material - mat_view);
        stageInteraction(stage_buf + (size_t)slot * kStageDoubles, q);
    }
    __device__ void stage(uint32_t slot, const InteractionT<false>& ia) const {
        if (stage_buf) stage_(slot, ia);
    }
    __device__ void request(uint32_t slot, bool want, bool global) const {
        const unsigned long long m = waveBallot(want);
        if (!m) return;
        const uint32_t lane = laneId();
        const int leader = __ffsll((long long)waveBallot(true)) - 1;
        unsigned long long base = 0ull;
        if ((int)lane == leader) base = atomicAdd(rcount, (unsigned long long)__popcll(m));
        base = waveBroadcast64(base, leader);
        if (want) requests[base + __popcll(m & ((1ull << lane) - 1ull))] = slot | (global ? 0x80000000u : 0u);
    }
    __device__ bool any(bool b) const { return waveBallot(b) != 0ull; }
    __device__ unsigned long long pop(bool need) const { return wavePop(need, work); }
    __device__ void filmAdd(double* a, double v) const { atomicAdd(a, v); }  // std::atomic<double> of Film::Splat
    __device__ void prevRay(uint32_t entry, d3& o, d3& d) const { queue.prevRay(entry, o, d); }
    __device__ uint32_t push(uint32_t slot, bool p0, bool p1, d3 o0, d3 d0, d3 o1, d3 d1, double dist1, uint32_t light1) const {
        const unsigned long long m0 = waveBallot(p0), m1 = waveBallot(p1);
        const uint32_t n0 = __popcll(m0), total = n0 + __popcll(m1);
        if (!total) return 0u;
        const uint32_t lane = laneId();
        const int leader = __ffsll((long long)waveBallot(true)) - 1;
        unsigned long long base = 0ull;
        if ((int)lane == leader) base = atomicAdd(count, (unsigned long long)total);
        base = waveBroadcast64(base, leader);
        const unsigned long long below = (1ull << lane) - 1ull;
        // the wave's bounce rays first, then its shadow rays (neighbouring queue entries = similar rays)
        const uint32_t e0 = (uint32_t)base + (uint32_t)__popcll(m0 & below);
        if (p0) queue.put(e0, slot * 2u, o0, d0, 0.0, kNoSurface);
        if (p1) queue.put(base + n0 + __popcll(m1 & below), slot * 2u + 1u, o1, d1, dist1, light1);
        return e0;
    }
};

// 3 waves per SIMD (168 VGPRs, 148 B/lane of scratch): the launch is bound by the latency of its scattered scene reads;
// measured C3 1122 -> 1158 Mray/s against the natural 221 VGPRs / 2 waves, 1017 at 4 waves (spills take over).
template <bool kPhoton>
__global__ void __launch_bounds__(kWfBlock) __attribute__((amdgpu_waves_per_eu(3, 3))) wfShadeKernel(const DeviceScene scene, const WfShadeArgs a) {
    MCRT_DYNAMIC_LDS(lds, 16);
    MCRT_LDS_AS uint32_t* ltab = ldsAt<uint32_t>(lds, 0);
    // Staging in ONE round trip: every load of the Sobol tables, the sin/cos table and the lane's own flags word is issued before
    // the first LDS write (loop after loop, each waiting for its loads, was five to six trips before a workgroup could start).
    static_assert(kWfBlock == 256 && kSobolTableWords == 6 * 4 * 256 && sizeof(glibc235::kSinCosTab) == 440 * 8, "staging below is written for these sizes");
    const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = local < a.slot_count;
    const uint32_t slot = a.slot_base + (valid ? local : 0u);
    unsigned long long fw;
    {
        fw = wfSlotFlags(a.pool, slot, valid);
        const uint4* s4 = reinterpret_cast<const uint4*>(scene.sobol_tab);
        uint4 sob[6];
#pragma unroll
        for (int k = 0; k < 6; k++) sob[k] = s4[threadIdx.x + k * 256];
        const unsigned long long sc0 = glibc235::kSinCosTab[threadIdx.x];
        const unsigned long long sc1 = glibc235::kSinCosTab[threadIdx.x + 256u < 440u ? threadIdx.x + 256u : 0u];
        MCRT_LDS_AS uint4* l4 = reinterpret_cast<MCRT_LDS_AS uint4*>(ltab);
#pragma unroll
        for (int k = 0; k < 6; k++) l4[threadIdx.x + k * 256] = sob[k];
        glibc235::ldsTabStore(threadIdx.x, sc0);
        if (threadIdx.x + 256u < 440u) glibc235::ldsTabStore(threadIdx.x + 256u, sc1);
    }
    RefractionHistory rh;
    rh.iors = ldsAt<double>(lds, kSobolTableWords * 4u) + threadIdx.x;
    rh.stride = blockDim.x;
    rh.size = 0;
    ShadeViewT<false> sh;
    sh.num_lights = scene.num_lights;
    sh.scene_ior = scene.scene_ior;
    sh.surf_v = scene.surf_v;
    sh.surf_normal = scene.surf_normal;
    sh.surf_rec = scene.surf_rec;
    sh.surf_vn = scene.surf_vn;
    sh.surf_area = scene.surf_area;
    sh.surf_material = scene.surf_material;
    sh.surf_kind = scene.surf_kind;
    sh.materials = scene.materials;
    sh.light_surface = scene.light_surface;
    sh.light_cdf = scene.light_cdf;
    if (a.lds_tables) {
        unsigned char* base = lds + kSobolTableWords * 4u + kMaxIors * kWfBlock * 8u;
        uint64_t* lmat = reinterpret_cast<uint64_t*>(base);
        double* lcdf = reinterpret_cast<double*>(base + alignUp(scene.num_materials * (uint32_t)sizeof(mcrt_material), 16));
        uint32_t* lsurf = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(lcdf) + alignUp(scene.num_lights * 8u, 16));
        const uint64_t* gmat = reinterpret_cast<const uint64_t*>(scene.materials);
        const uint32_t mwords = scene.num_materials * (uint32_t)(sizeof(mcrt_material) / 8), nl = scene.num_lights;
        // (the first two words per lane of the materials and the first light entry are asked for together: one trip for the usual sizes)
        const uint32_t t = threadIdx.x;
        const uint64_t m0 = gmat[t < mwords ? t : 0u], m1 = gmat[t + 256u < mwords ? t + 256u : 0u];
        const double c0 = nl ? scene.light_cdf[t < nl ? t : 0u] : 0.0;
        const uint32_t s0 = nl ? scene.light_surface[t < nl ? t : 0u] : 0u;
        if (t < mwords) lmat[t] = m0;
        if (t + 256u < mwords) lmat[t + 256u] = m1;
        if (t < nl) {
            lcdf[t] = c0;
            lsurf[t] = s0;
        }
        for (uint32_t i = t + 512u; i < mwords; i += blockDim.x) lmat[i] = gmat[i];
        for (uint32_t i = t + 256u; i < nl; i += blockDim.x) {
            lcdf[i] = scene.light_cdf[i];
            lsurf[i] = scene.light_surface[i];
        }
        sh.materials = reinterpret_cast<const mcrt_material*>(lmat);
        sh.light_cdf = lcdf;
        sh.light_surface = lsurf;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *a.count_reset = 0ull;
        *a.pop_reset = 0ull;
        if (kPhoton) {
            *a.rcount_reset = 0ull;
            *a.rpop_reset = 0ull;
        }
    }
    __syncthreads();
    DevWfEnv env{a.work, a.queue, a.count_out, a.requests, a.rcount_out, a.stage, sh.materials, scene.materials};
    uint32_t paths = 0;
    wfShadeSlot<false, kPhoton>(env, a.pool, slot, fw, a.fr, sh, rh, (SobolTab)ltab, paths, &a.pm);
    waveAccumulate(a.stats + 0, paths);
    waveAccumulate(a.stats + 7, rh.overflow ? 1u : 0u);
}

// The per-sample store of renderKernel / renderKernelSM -> image: rgb_sum += radiance * 1 in sample order (Film::deposit /
// Splat::update, film.cpp:61-79,99-105), then Splat::get = max(sum / weight_sum, 0) (film.cpp:107-113). One lane per pixel;
// consecutive lanes read consecutive 24-byte records of a sample plane.
__global__ void __launch_bounds__(256) sampleResolveKernel(const double* samples, uint64_t pass_pixels, uint32_t spp, double* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pass_pixels) return;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    const double* p = samples + i * 3;
    for (uint32_t s = 0; s < spp; s++, p += pass_pixels * 3) {
        acc0 += p[0];
        acc1 += p[1];
        acc2 += p[2];
    }
    const double wsum = (double)spp;
    out[i * 3] = gmax(acc0 / wsum, 0.0);
    out[i * 3 + 1] = gmax(acc1 / wsum, 0.0);
    out[i * 3 + 2] = gmax(acc2 / wsum, 0.0);
}

// Film::scan over the frame (film.cpp:81-84,107-113): the splats of a reconstruction-filter frame -> image.
__global__ void filmResolveKernel(const double* blob, uint64_t pixels, double* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pixels) filmResolve(blob + i * 4, out + i * 3);
}

// kNN launch of the wavefront photon mapper: one estimate request per wave at a time (mcrt_waveknn.hpp), the k photons
// of every search written out for the next shade launch. 80 VGPRs: 6 waves per SIMD.
struct WfKnnArgs {
    WfPool pool;
    const uint32_t* requests;
    const unsigned long long* count;
    unsigned long long* pop;
    unsigned long long* stats;
    PhotonMapViewW maps[2];  // global, caustic
    uint32_t k;
    uint32_t* res_n;
    double* res_r2;
    uint32_t* res_idx;
    double* res_d2;
    const double* stage;  // [slots][kStageDoubles] the requests' Interactions (written by the shade launch); null: write the photons out
    double* est;          // [slots][6] the estimates, caustic rgb then global rgb
    uint32_t* spill;      // [waves][kWaveSpill][3] frontier entries beyond the registers' (mcrt_waveknn.hpp)
};

// kEval: the launch evaluates the estimate itself — the k photons' BSDF terms by k lanes at once, a wave reduction
// (waveEvalPhotons, as in renderKernelPM) — from the Interaction the shade launch staged, instead of handing the k photons
// back for a per-lane loop in the next shade launch (2 x k x 12 B per slot written and read, k divergent BSDF evaluations).
// Occupancy of the kNN launch (round 6, C5 through the pipeline at 64 spp, ms per frame; profiles/r06_ab_knn_occupancy.log): the
// compiler's own choice (165 VGPRs, 3 waves per SIMD, no spill) 882.9; 4 waves (128 VGPRs, 21 spilled) 828.0; 5 waves (96 VGPRs, 61
// spilled) 816.0 - a search is a chain of dependent reads (octree records, leaf runs), and waves are what hides it.
#ifndef MCRT_KNN_OCC
#define MCRT_KNN_OCC __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
template <bool kEval, int R = kWaveRows>
__global__ void __launch_bounds__(256) MCRT_KNN_OCC wfKnnKernel(const WfKnnArgs a) {
    __shared__ double s_d2[4 * waveCand(R)];
    __shared__ uint32_t s_idx[4 * waveCand(R)];
    __shared__ uint32_t s_hist[4 * kWaveHist];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    WaveKnnLds W;
    W.d2 = (MCRT_LDS_AS double*)s_d2 + wave * waveCand(R);
    W.idx = (MCRT_LDS_AS uint32_t*)s_idx + wave * waveCand(R);
    W.hist = (MCRT_LDS_AS uint32_t*)s_hist + wave * kWaveHist;
    W.spill = a.spill + ((size_t)blockIdx.x * 4u + wave) * (3u * kWaveSpill);
    const unsigned long long n = *a.count;
    const uint32_t slots = a.pool.n;
    uint32_t overflow = 0, visits = 0, searches = 0;
    // Requests are taken kKnnBatch at a time (round 6). One returning atomic per REQUEST on one word was this launch's ceiling: a
    // device-scope counter serves ~88 M dequeues per second (MI355X_MICROARCH.md, price list "dequeue") and C5 files 126 M requests per
    // 64-spp frame - the launch ran at 110 M searches/s whatever the searches cost (profiles/r06_kernel_trace_c5_pipeline_before.md).
    // Now lane j of the wave reads request base + j and that slot's position (three coalesced round trips for the whole batch instead of
    // three dependent ones per request), and the wave works the batch off with readlane broadcasts.
    constexpr uint32_t kKnnBatch = 16;
    for (;;) {
        unsigned long long w0 = 0ull;
        if (lane == 0) w0 = atomicAdd(a.pop, (unsigned long long)kKnnBatch);
        w0 = waveBroadcast64(w0, 0);
        if (w0 >= n) break;
        const uint32_t in_batch = n - w0 < (unsigned long long)kKnnBatch ? (uint32_t)(n - w0) : kKnnBatch;
        uint32_t req_l = 0u;
        d3 p_l = splat(0.0);
        if (lane < in_batch) {
            req_l = a.requests[w0 + lane];
            const uint32_t s = req_l & 0x7FFFFFFFu;
            // Interaction::position = ray(t) (interaction.cpp:15)
            p_l = a.pool.get3(kWfRayO, s) + a.pool.get3(kWfRayD, s) * a.pool.getd(kWfHit0T, s);
        }
        for (uint32_t j = 0; j < in_batch; j++) {
        const uint32_t req = (uint32_t)__builtin_amdgcn_readlane((int)req_l, (int)j);
        const uint32_t slot = req & 0x7FFFFFFFu;
        const d3 p = d3{bitsD(waveBroadcast64(dBits(p_l.x), (int)j)), bitsD(waveBroadcast64(dBits(p_l.y), (int)j)), bitsD(waveBroadcast64(dBits(p_l.z), (int)j))};
        for (int map = 1; map >= ((req >> 31) ? 0 : 1); map--) {  // caustic map always, global map on request
            double r2;
            const uint32_t c = waveKnnSearch<R>(a.maps[map], p, a.k, W, r2, overflow, visits);
            searches++;
            if constexpr (kEval) {
                d3 sum = splat(0.0);
                if (c) {  // else photons.empty(): the estimate is zero (photon-mapper.cpp:347, :374)
                    InteractionT<false> q;
                    loadStagedInteraction(a.stage + (size_t)slot * kStageDoubles, q);
                    sum = waveEvalPhotons(q, a.maps[map], map == 1, W.d2, W.idx, c, r2);
                }
                if (lane == 0) {
                    double* o = a.est + (size_t)slot * 6 + (map == 1 ? 0 : 3);
                    o[0] = sum.x;
                    o[1] = sum.y;
                    o[2] = sum.z;
                }
                continue;
            }
            if (lane == 0) {
                a.res_n[(size_t)map * slots + slot] = c;
                a.res_r2[(size_t)map * slots + slot] = r2;
            }
            for (uint32_t j2 = lane; j2 < c; j2 += 64) {
                const size_t at = ((size_t)map * a.k + j2) * slots + slot;
                a.res_idx[at] = W.idx[j2];
                a.res_d2[at] = W.d2[j2];
            }
        }
        }
    }
    if (lane == 0) {
        if (searches) atomicAdd(a.stats + 4, (unsigned long long)searches);
        if (visits) atomicAdd(a.stats + 6, (unsigned long long)visits);
        if (overflow) atomicAdd(a.stats + 5, kKnnOverflowUnit);
    }
}

// ------------------------------------------------------------------------------------------------
// photon-mapping eye pass with wave-cooperative radiance estimates (mcrt_waveknn.hpp)
// ------------------------------------------------------------------------------------------------
// (traceWalkQ / traceWalkShared: the wave-synchronous walks over the quantised child blocks, mcrt_sharedleaf.hpp)
// What traceWalkQ needs, carved out of the LDS plan of the wave-synchronous kernels (planLds): the top child blocks take
// the place of the staged node records, the traversal stack region is used with the state machine's 8-byte entries.
struct QWalk {
    SmSceneView<false> sv;
    QView<true> qv;
    SmStack stk;
};
__device__ inline void setupQWalk(const DeviceScene& scene, unsigned char* lds, const LaneStack& lstk, QWalk& q, uint32_t stack_depth,
                                  bool with_iors = true) {  // every thread calls
    const LdsPlan lp = planLds(scene, blockDim.x, true, stack_depth, with_iors ? (uint32_t)kMaxIors : kPmLdsIors);
    MCRT_LDS_AS QBlock* lq = ldsAt<QBlock>(lds, lp.node_bounds);
    const uint32_t room = scene.stage_nodes * 56u / 64u;
    q.qv.blocks = scene.qblocks;
    q.qv.lds_blocks = room < scene.num_qblocks ? room : scene.num_qblocks;
    q.qv.lds_ptr = lq;
    q.qv.root_a = scene.q_root_a;
    q.qv.root_m = scene.q_root_m;
    __syncthreads();  // setupViews' copies of the node records are not read by these kernels
    for (uint32_t i = threadIdx.x; i < q.qv.lds_blocks * 16u; i += blockDim.x)
        reinterpret_cast<MCRT_LDS_AS uint32_t*>(lq)[i] = reinterpret_cast<const uint32_t*>(scene.qblocks)[i];
    __syncthreads();
    q.sv.num_nodes = scene.num_nodes;
    q.sv.nodes = scene.nodes64;
    q.sv.prim = scene.prim;
    q.sv.lds_nodes = 0;
    q.sv.lds_node_ptr = nullptr;
    q.stk.lds = reinterpret_cast<MCRT_LDS_AS SmStackEntry*>(lstk.lds);  // same 8-byte entries, same [depth][lanes] region
    q.stk.lds_depth = (int)stack_depth;
    q.stk.lds_stride = lstk.lds_stride;
    q.stk.spill = reinterpret_cast<SmStackEntry*>(lstk.spill);
    q.stk.spill_stride = lstk.spill_stride;
    q.stk.max_depth = lstk.max_depth;
}

struct PmExtra {
    PhotonMapViewW global_map, caustic_map;
    uint32_t stack_depth;  // traversal-stack entries per lane kept in LDS (tree in HBM: the state machine's stack, any depth; else kLdsStackDepth)
    double* iors_global;   // refraction-history entries beyond the first kPmLdsIors, [kMaxIors - kPmLdsIors][lanes] (1024-lane instance), or null: all in LDS
    double* stage;         // estimate requests, [lanes][kStageDoubles]: what Interaction::BSDF reads of a lane's Interaction (mcrt_waveknn.hpp)
    uint32_t* knn_spill;   // [waves][kWaveSpill][3]: the searches' frontier entries beyond the registers' 128 (mcrt_waveknn.hpp)
};

// kLanes: 512 (2 waves per SIMD, 256 VGPRs) or 1024 (4 waves per SIMD, 128 VGPRs). The kernel spends 97.6 % of its wave
// cycles inside the radiance estimates (measured, hexagon_room maps), whose throughput follows the resident waves
// (knnWaveKernel alone: 61 / 112 / 172 / 188 M searches/s at 1 / 2 / 4 / 5 waves per SIMD) — the spills the narrow
// register budget causes in the per-lane path code do not matter next to that.
template <bool kCount, bool kAll, int kLanes = (int)kBlock, int R = kWaveRows>
__global__ void __launch_bounds__(kLanes) renderKernelPM(const DeviceScene scene, const RenderParams prm, const PmExtra pmx) {
    MCRT_DYNAMIC_LDS(lds, 16);
    SceneViewT<kAll> sv;
    ShadeViewT<kAll> sh;
    SobolTab tab;
    LaneStack stk;
    RefractionHistory rh;
    setupViews<kAll>(scene, lds, sv, sh, tab, stk, rh, prm.spill, prm.total_lanes, pmx.stack_depth, pmx.iors_global);
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    // tree in HBM: walk it through the quantised child blocks
    QWalk qw;
    if constexpr (!kAll) setupQWalk(scene, lds, stk, qw, pmx.stack_depth, pmx.iors_global == nullptr);
    auto intersect = [&](const Ray& ray, bool shadow, const ShadowQuery* sq) {
        if constexpr (kAll) {
            return shadow ? sceneIntersect<kAll, kCount, true>(sv, ray, stk, cnt, sq) : sceneIntersect<kAll, kCount, false>(sv, ray, stk, cnt);
        } else {
            return traceWalkQ<kCount>(qw.sv, qw.qv, qw.stk, ray, shadow, sq, cnt);
        }
    };
    // per-wave candidate buffer behind the common LDS plan
    WaveKnnLds W;
    {
        const uint32_t base = alignUp(planLds(scene, blockDim.x, true, pmx.stack_depth, pmx.iors_global ? kPmLdsIors : (uint32_t)kMaxIors).total, 16);
        const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
        W.d2 = ldsAt<double>(lds, base) + wave * waveCand(R);
        W.idx = ldsAt<uint32_t>(lds, base + waves * waveCand(R) * 8u) + wave * waveCand(R);
        W.hist = ldsAt<uint32_t>(lds, base + waves * waveCand(R) * 12u) + wave * kWaveHist;
        W.spill = pmx.knn_spill + ((size_t)blockIdx.x * waves + wave) * (3u * kWaveSpill);
        if constexpr (!kAll) {  // (trees in memory: the list's state in LDS, behind all the waves' buffers - mcrt_waveknn.hpp)
            W.state = ldsAt<uint32_t>(lds, base + waves * waveKnnBytes(R)) + wave * (kWaveStateBytes / 4u);
            waveKnnInit(W, W.spill);
        }
    }

    double* const stage_lane = pmx.stage + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * kStageDoubles;
    double* const stage_wave = pmx.stage + ((size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * kStageDoubles;

    PathState st;
    uint32_t paths = 0, searches = 0, octant_visits = 0, knn_overflow = 0;
    bool have_pixel = false, path_active = false, exhausted = false;
    uint32_t px = 0, py = 0, ly = 0, sample = 0, sample_end = 0;
    const uint32_t W_img = prm.cam.width;
    const bool direct_visualization = prm.direct_visualization != 0;
    unsigned long long cyc_est = 0ull;  // MCRT_COUNT_TESTS: wave cycles inside the radiance estimates / in the kernel
    const unsigned long long cyc_begin = kCount ? clock64() : 0ull;

    for (;;) {
        const bool need = !have_pixel && !exhausted;
        if (waveBallot(need)) {
            const unsigned long long w = wavePop(need, prm.work_counter);
            if (need) {
                if (w >= prm.work_items) {
                    exhausted = true;
                } else {
                    const WorkUnit u = decodeUnit(prm, w);
                    if (u.lx < W_img && u.ly < prm.row_end && u.first < u.end) {
                        px = u.lx;
                        ly = u.ly;
                        py = localToGlobalRow(prm.cam, ly);
                        have_pixel = true;
                        sample = u.first;
                        sample_end = u.end;
                        st.smp.initiate(prm.global_seed, py * W_img + px);
                    }
                }
            }
        }
        if (!waveBallot(have_pixel)) {
            if (!waveBallot(!exhausted)) break;
            continue;
        }
        if (have_pixel && !path_active) {
            st.smp.setIndex(sample);
            pathBegin(st, rh, cameraRay<!kAll>(prm.cam, sh.scene_ior, px, py, st.smp, tab));
            path_active = true;
            paths++;
        }

        // ---- part 1 (per lane): PhotonMapper::sampleRay up to the radiance estimates (photon-mapper.cpp:288-313)
        InteractionT<kAll> ia;
        ia.material = sh.materials;
        ia.position = ia.out = ia.shading_cs.c0 = ia.shading_cs.c1 = ia.shading_cs.c2 = splat(0.0);
        ia.n1 = ia.n2 = ia.R = ia.T = 0.0;
        ia.type = kDiffuse;
        ia.inside = false;
        ia.dirac_delta = false;
        bool ended = false, needC = false, needG = false;
        Hit walked;  // (tree in memory: the walk is a whole-wave affair, traceWalkShared - every lane calls it)
        if constexpr (!kAll) walked = traceWalkShared<kCount>(qw.sv, qw.qv, qw.stk, path_active, st.ray, false, nullptr, cnt, reinterpret_cast<MCRT_LDS_AS uint8_t*>(W.hist));
        if (path_active) {
            st.smp.shuffle();
            Hit isect;
            if constexpr (kAll) isect = intersect(st.ray, false, nullptr);
            else isect = walked;
            if (isect.surface == kNoSurface) {
                ended = true;  // no sky in photon mode (:292-295)
            } else {
                interactionInit(ia, sh, isect, st.ray, rh.externalIOR(st.ray), st.smp, tab);
                st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;
                if (ia.dirac_delta) {
                    if (!st.ray.dirac_delta && st.ray.depth != 0) ended = true;  // :303-306
                } else {
                    needC = true;                                                                       // :315
                    needG = !(!direct_visualization && (st.ray.dirac_delta || st.ray.depth == 0));     // :317 / :327
                }
            }
        }
        // ---- part 2 (whole wave): caustic estimates, then global estimates
        const unsigned long long t_est = kCount ? clock64() : 0ull;
        if (needC) stageInteraction(stage_lane, ia);  // needG implies needC
        __threadfence_block();                        // the records are read by the other lanes of the wave
        const d3 C = waveEstimate<kAll, R>(needC, stage_wave, pmx.caustic_map, prm.k_nearest, true, W, searches, octant_visits, knn_overflow);
        if (needC) st.radiance = st.radiance + C * st.throughput;
        const d3 G = waveEstimate<kAll, R>(needG, stage_wave, pmx.global_map, prm.k_nearest, false, W, searches, octant_visits, knn_overflow);
        if (needG) {
            st.radiance = st.radiance + G * st.throughput;  // :330, the path ends here
            ended = true;
        }
        if (kCount) cyc_est += clock64() - t_est;
        // ---- part 3 (per lane): next-event estimate, BSDF sampling, russian roulette (:308-311, :319-325, :334-339)
        DirectQuery dq;
        bool want_shadow = false;
        if (path_active && !ended && !ia.dirac_delta) want_shadow = sampleDirectSetup(sh, ia, st.ls, dq, st.smp, tab);
        if constexpr (kAll) {
            if (want_shadow) {
                Hit shadow = intersect(dq.shadow_ray, true, &dq.sq);
                st.radiance = st.radiance + sampleDirectFinish(sh, ia, st.ls, dq, shadow) * st.throughput;
            }
        } else {
            const Hit shadow = traceWalkShared<kCount>(qw.sv, qw.qv, qw.stk, want_shadow, dq.shadow_ray, true, &dq.sq, cnt, reinterpret_cast<MCRT_LDS_AS uint8_t*>(W.hist));
            if (want_shadow) st.radiance = st.radiance + sampleDirectFinish(sh, ia, st.ls, dq, shadow) * st.throughput;
        }
        if (path_active && !ended) {
            d3 bsdf_absIdotN;
            if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) {
                ended = true;
            } else {
                st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);
                if (absorb(st.ray, st.throughput, st.smp, tab)) ended = true;
                else rh.update(st.ray);
            }
        }
        if (path_active && ended) {
            storeSample(prm, sample, px, ly, st.radiance);
            path_active = false;
            if (++sample == sample_end) have_pixel = false;
        }
    }
    waveAccumulate(prm.stats + 0, paths);
    waveAccumulate(prm.stats + 1, cnt.rays);
    if (kCount) {
        waveAccumulate(prm.stats + 2, cnt.node_tests);
        waveAccumulate(prm.stats + 3, cnt.prim_tests);
    }
    waveAccumulate(prm.stats + 4, searches);
    // (one word for both overflows, as in round 4: lanes whose traversal stack overflowed count 1 each, a search whose frontier overflowed
    // sets kKnnOverflowFlag - the host reads the bits above 15. Two separate additions here cost C5 9 % of a frame: this kernel's
    // register allocation - 128 VGPRs, ~700 spilled - turns on such things, profiles/r05_ab_c5_bisect.log)
    waveAccumulate(prm.stats + 5, cnt.overflow | knn_overflow);
    waveAccumulate(prm.stats + 7, rh.overflow ? 1u : 0u);
    waveAccumulate(prm.stats + 6, octant_visits);
    if (kCount && __lane_id() == 0) {
        atomicAdd(prm.stats + 8, cyc_est);
        atomicAdd(prm.stats + 9, (unsigned long long)(clock64() - cyc_begin));
    }
}

// LinearOctree::knnSearch operator: one query at a time per wave
template <int R = kWaveRows>
__global__ void __launch_bounds__(256) knnWaveKernel(const PhotonMapViewW map, uint64_t n, const double* p, uint32_t k, uint32_t* out_count,
                                                     uint32_t* out_index, double* out_d2, unsigned long long* flags, uint32_t* spill) {
    __shared__ double s_d2[4 * waveCand(R)];
    __shared__ uint32_t s_idx[4 * waveCand(R)];
    __shared__ uint32_t s_hist[4 * kWaveHist];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    WaveKnnLds W;
    W.d2 = (MCRT_LDS_AS double*)s_d2 + wave * waveCand(R);
    W.idx = (MCRT_LDS_AS uint32_t*)s_idx + wave * waveCand(R);
    W.hist = (MCRT_LDS_AS uint32_t*)s_hist + wave * kWaveHist;
    W.spill = spill + ((size_t)blockIdx.x * 4u + wave) * (3u * kWaveSpill);
    const uint64_t waves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t overflow = 0, visits = 0;
    for (uint64_t q = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave; q < n; q += waves_total) {
        const d3 pt = ld3(p + 3 * q);
        double r2;
        const uint32_t c = waveKnnSearch<R>(map, pt, k, W, r2, overflow, visits);
        waveSortResult<(R <= 4 ? 2 : R - 4)>(W, c);
        if (lane == 0) out_count[q] = c;
        for (uint32_t j = lane; j < k; j += 64) {
            out_index[q * k + j] = j < c ? W.idx[j] : 0xFFFFFFFFu;
            out_d2[q * k + j] = j < c ? W.d2[j] : INFINITY;
        }
    }
    if (overflow && lane == 0) atomicAdd(flags, 1ull);
}

// LinearOctree::knnSearch operator, four queries at a time per wave (mcrt_groupknn.hpp); a row that gives up has its query
// repeated by the whole wave
__global__ void __launch_bounds__(256) knnGroupKernel(const PhotonMapViewW map, uint64_t n, const double* p, uint32_t k, uint32_t* out_count,
                                                      uint32_t* out_index, double* out_d2, unsigned long long* flags) {
    __shared__ __align__(16) unsigned char s_buf[4 * kGroupKnnBytes];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, group = lane >> 4, l16 = lane & 15u;
    MCRT_LDS_AS unsigned char* wb = (MCRT_LDS_AS unsigned char*)s_buf + wave * kGroupKnnBytes;
    const GroupKnnLds G = groupKnnLds(wb, group);
    const WaveKnnLds W = waveKnnLdsOver(wb);
    const uint64_t waves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t overflow = 0, visits = 0;
    for (uint64_t q0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 4u; q0 < n; q0 += waves_total * 4u) {
        const uint64_t q = q0 + group;
        const bool on = q < n;
        const d3 pt = on ? ld3(p + 3 * q) : splat(0.0);
        uint32_t c = 0;
        double r2 = 0.0;
        bool redo = false;
        groupKnnSearch(map, pt, on, k, G, c, r2, redo, visits);
        // the row's result ascending by (distance2, index): ranks by counting, written straight to the output
        __builtin_amdgcn_wave_barrier();
        {
            double my_d[4];
            uint32_t my_i[4], rank[4];
            for (int s = 0; s < 4; s++) {
                const uint32_t j = l16 + 16u * s;
                const bool v = on && !redo && j < c;
                my_d[s] = v ? G.d2[j] : INFINITY;
                my_i[s] = v ? G.idx[j] : 0xFFFFFFFFu;
                rank[s] = 0;
            }
            for (uint32_t i = 0; waveBallot(on && !redo && i < c); i++) {
                const bool v = on && !redo && i < c;
                const double d = v ? G.d2[i] : INFINITY;
                const uint32_t id = v ? G.idx[i] : 0xFFFFFFFFu;
                for (int s = 0; s < 4; s++) rank[s] += (d < my_d[s] || (d == my_d[s] && id < my_i[s])) ? 1u : 0u;
            }
            if (on && !redo) {
                if (l16 == 0) out_count[q] = c;
                for (int s = 0; s < 4; s++) {
                    const uint32_t j = l16 + 16u * s;
                    if (j < c) {
                        out_index[q * k + rank[s]] = my_i[s];
                        out_d2[q * k + rank[s]] = my_d[s];
                    } else if (j < k) {
                        out_index[q * k + j] = 0xFFFFFFFFu;
                        out_d2[q * k + j] = INFINITY;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long rm = waveBallot(on && redo && l16 == 0u);
        while (rm) {  // (rare) rows that gave up: their queries one at a time by the whole wave
            const int src = __ffsll((long long)rm) - 1;
            rm &= rm - 1;
            const uint64_t qq = q0 + ((uint32_t)src >> 4);
            const d3 qp = waveShfl3(pt, src);
            double rr;
            const uint32_t cc = waveKnnSearch(map, qp, k, W, rr, overflow, visits);
            waveSortResult(W, cc);
            if (lane == 0) out_count[qq] = cc;
            for (uint32_t j = lane; j < k; j += 64) {
                out_index[qq * k + j] = j < cc ? W.idx[j] : 0xFFFFFFFFu;
                out_d2[qq * k + j] = j < cc ? W.d2[j] : INFINITY;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (overflow && lane == 0) atomicAdd(flags, 1ull);
}

// ------------------------------------------------------------------------------------------------
// photon emission pass (§8(f) rank 1): one photon path per lane at a time, regenerated like the eye paths
// ------------------------------------------------------------------------------------------------
struct EmitParams {
    uint32_t num_lights;
    const unsigned long long* light_first;  // [num_lights + 1] prefix sums of the per-light emission counts
    const double* light_photon_flux;        // [num_lights][3]
    unsigned long long total_emissions;     // this launch handles paths [first_emission, total_emissions)
    unsigned long long first_emission;
    uint32_t stride;                        // 1; the sizing pilot takes every stride-th path of the range (capacity 0: it only counts)
    uint32_t global_seed;
    double non_caustic_reject;
    float* photons[2];                      // 0 global, 1 caustic: [capacity][8]
    unsigned long long* keys[2];
    unsigned long long capacity[2];
    unsigned long long* counters;           // [0] work, [1] global count, [2] caustic count, [3] paths, [4] rays, [5] overflow
    StackEntry* spill;
    uint32_t total_lanes;
};

// wave-aggregated append: lanes with store == true get consecutive slots of the list
__device__ inline unsigned long long waveAppend(bool store, unsigned long long* counter) { return wavePop(store, counter); }

template <bool kAll>
__global__ void __launch_bounds__(kBlock) emitKernel(const DeviceScene scene, const EmitParams prm) {
    MCRT_DYNAMIC_LDS(lds, 16);
    SceneViewT<kAll> sv;
    ShadeViewT<kAll> sh;
    SobolTab tab;
    LaneStack stk;
    RefractionHistory rh;
    setupViews<kAll>(scene, lds, sv, sh, tab, stk, rh, prm.spill, prm.total_lanes);
    QWalk qw;  // tree in HBM: quantised child blocks
    if constexpr (!kAll) setupQWalk(scene, lds, stk, qw, (uint32_t)kLdsStackDepth);

    EmitState es;
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    uint32_t paths = 0;
    bool active = false, exhausted = false;
    for (;;) {
        const bool need = !active && !exhausted;
        if (waveBallot(need)) {
            const unsigned long long e = prm.first_emission + wavePop(need, prm.counters + 0) * prm.stride;
            if (need) {
                if (e >= prm.total_emissions) {
                    exhausted = true;
                } else {
                    // which light: largest i with light_first[i] <= e
                    uint32_t lo = 0, hi = prm.num_lights - 1;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) / 2;
                        if (prm.light_first[mid] <= e) lo = mid;
                        else hi = mid - 1;
                    }
                    const d3 pf = ld3(prm.light_photon_flux + 3 * (size_t)lo);
                    emitBegin(es, rh, sh, lo, (uint32_t)(e - prm.light_first[lo]), pf, prm.global_seed, tab);
                    active = true;
                    paths++;
                }
            }
        }
        if (!waveBallot(active)) {
            if (!waveBallot(!exhausted)) break;
            continue;
        }
        PhotonOut out;
        out.store = false;
        out.caustic = false;
        if (active) {
            bool done;
            if constexpr (kAll) {
                done = emitBounce<false, kAll>(es, rh, sv, sh, stk, cnt, tab, prm.non_caustic_reject, out);
            } else {
                es.smp.shuffle();  // photon-mapper.cpp:233 (emitBounce's first line)
                const Hit isect = traceWalkQ<false>(qw.sv, qw.qv, qw.stk, es.ray, false, nullptr, cnt);
                done = emitAfterHit(es, rh, sh, isect, tab, prm.non_caustic_reject, out);
            }
            if (done) active = false;
        }
        for (int which = 0; which < 2; which++) {
            const bool mine = out.store && (out.caustic == (which == 1));
            if (waveBallot(mine)) {
                const unsigned long long slot = waveAppend(mine, prm.counters + 1 + which);
                if (mine && slot < prm.capacity[which]) {
                    float* o = prm.photons[which] + slot * 8ull;
                    for (int k = 0; k < 8; k++) o[k] = out.rec[k];
                    prm.keys[which][slot] = out.key;
                }
            }
        }
    }
    waveAccumulate(prm.counters + 3, paths);
    waveAccumulate(prm.counters + 4, cnt.rays);
    waveAccumulate(prm.counters + 5, cnt.overflow);
    waveAccumulate(prm.counters + 6, rh.overflow ? 1u : 0u);  // a photon path nested deeper than the kMaxIors media a lane keeps (mcrt_emit_photons*: an error, not a silent wrong medium)
}

// ------------------------------------------------------------------------------------------------
// operator-level kernels
// ------------------------------------------------------------------------------------------------
template <bool kAll>
__global__ void __launch_bounds__(kBlock) intersectKernel(const DeviceScene scene, uint64_t n, const double* start,
                                                       const double* direction, double* out_t, uint32_t* out_surface,
                                                       double* out_uv, StackEntry* spill, uint32_t total_lanes) {
    MCRT_DYNAMIC_LDS(lds, 16);
    SceneViewT<kAll> sv;
    ShadeViewT<kAll> sh;
    SobolTab tab;
    LaneStack stk;
    RefractionHistory rh;
    setupViews<kAll>(scene, lds, sv, sh, tab, stk, rh, spill, total_lanes);
    TraceCounters cnt = {0u, 0u, 0u, 0u};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Ray ray = makeRay(ld3(start + 3 * i), ld3(direction + 3 * i), 1.0);
        Hit h = sceneIntersect<kAll, false, false>(sv, ray, stk, cnt);
        out_t[i] = h.t;
        out_surface[i] = h.surface;
        out_uv[2 * i] = h.u;
        out_uv[2 * i + 1] = h.v;
    }
}

__global__ void samplerKernel(const uint32_t* tab, uint64_t n, const uint32_t* pixel, const uint32_t* index,
                              uint32_t shuffles, uint32_t global_seed, double* out) {
    __shared__ uint32_t ltab[kSobolTableWords];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kSobolTableWords; i += blockDim.x) ltab[i] = tab[i];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Sampler s;
        s.initiate(global_seed, pixel[i]);
        s.setIndex(index[i]);
        for (uint32_t k = 0; k < shuffles; k++) s.shuffle();
        for (int d = 0; d < 7; d++) out[i * 7 + d] = s.get(d, (SobolTab)ltab);
    }
}

// mcrt_bsdf: the lobe functions behind Interaction::BSDF (ray/interaction.cpp:84-153) on caller-supplied local-frame
// directions — Fresnel::dielectric / conductor, GGX::reflection / transmission / visibleMicrofacet / D / Lambda and
// Material::diffuseReflection (Oren-Nayar) — one vector per lane, same device functions as the integrators.
// in[n][11] = wi(3) wo(3) n1 n2 alpha u v; out[n][18] (layout in include/mcrt.h).
// mcrt_libm: the restated libm functions (mcrt_libm.hpp) on arrays of arguments, one per lane.
__global__ void __launch_bounds__(256) libmKernel(int fn, uint64_t n, const double* a, const double* b, double* out0, double* out1) {
    glibc235::stageSinCosTab();
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double x = a[i];
        if (fn == 0) {
            double sn, cs;
            refSinCos(x, sn, cs);
            out0[i] = sn;
            out1[i] = cs;
        } else if (fn == 1) {
            out0[i] = refSin(x);
        } else if (fn == 2) {
            out0[i] = refCos(x);
        } else if (fn == 3) {
            out0[i] = refAsin(x);
        } else if (fn == 5) {
            float sn, cs;
            refSinCosF((float)x, sn, cs);
            out0[i] = (double)sn;
            out1[i] = (double)cs;
        } else {
            out0[i] = refAtan2(x, b[i]);
        }
    }
}

struct BsdfKatConsts {
    mcrt_material rough;   // roughness, reflectance, A, B, MCRT_MAT_ROUGH
    double real[3], imag[3];
};
__global__ void __launch_bounds__(256) bsdfKernel(uint64_t n, const double* in, const BsdfKatConsts c, double* out) {
    glibc235::stageSinCosTab();
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double* I = in + 11 * i;
        double* O = out + 18 * i;
        const d3 wi = ld3(I), wo = ld3(I + 3);
        const double n1 = I[6], n2 = I[7], al = I[8], u = I[9], v = I[10];
        double pdf;
        O[0] = fresnelDielectric(n1, n2, wo.z);
        const d3 fc = fresnelConductor(n1, ld3(c.real), ld3(c.imag), wo.z);
        O[1] = fc.x; O[2] = fc.y; O[3] = fc.z;
        d3 wir = wi;
        wir.z = fabs(wir.z) + 1e-3;
        wir = normalize(wir);
        O[4] = ggxReflection(wir, wo, al, al, pdf); O[5] = pdf;
        O[6] = ggxTransmission(-wir, wo, n1, n2, al, al, pdf); O[7] = pdf;
        const d3 m = ggxVisibleMicrofacet(u, v, wo, al, al);
        O[8] = m.x; O[9] = m.y; O[10] = m.z;
        O[11] = ggxD(m, al, al);
        O[12] = ggxLambda(wo, al, al);
        const d3 d = matDiffuseReflection(c.rough, wir, wo, pdf);
        O[13] = d.x; O[14] = d.y; O[15] = d.z; O[16] = pdf; O[17] = 0.0;
    }
}

__global__ void knnKernel(const PhotonMapView map, uint64_t n, const double* p, uint32_t k, uint32_t* out_count,
                          uint32_t* out_index, double* out_d2, double* res_d2, uint32_t* res_idx, double* visit_d2,
                          uint32_t* visit_oct, uint32_t total_lanes, uint32_t max_visit, unsigned long long* overflow_flag) {
    const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
    KnnScratch ks;
    ks.res_d2 = res_d2 + gl;
    ks.res_idx = res_idx + gl;
    ks.visit_d2 = visit_d2 + gl;
    ks.visit_oct = visit_oct + gl;
    ks.stride = total_lanes;
    ks.max_visit = max_visit;
    ks.overflowed = 0u;
    for (uint64_t i = gl; i < n; i += total_lanes) {
        uint32_t visits = 0;
        uint32_t c = knnSearch(map, ld3(p + 3 * i), k, ks, visits);
        out_count[i] = c;
        // heap-sort the result in place (ascending distance, ties by index) into the output row
        for (uint32_t q = 0; q < k; q++) {
            out_index[i * k + q] = 0xFFFFFFFFu;
            out_d2[i * k + q] = INFINITY;
        }
        // selection by repeated extraction of the max-heap root
        uint32_t size = c;
        while (size > 0) {
            KnnEntry top = ks.res(0);
            out_index[i * k + (size - 1)] = top.index;
            out_d2[i * k + (size - 1)] = top.distance2;
            KnnEntry last = ks.res(size - 1);
            size--;
            if (size > 0) knnSiftDown(ks, size, last, 0);
        }
        // fix tie order (equal distance2: ascending index) with a local insertion pass
        for (uint32_t a = 1; a < c; a++) {
            uint32_t b = a;
            while (b > 0 && out_d2[i * k + b - 1] == out_d2[i * k + b] && out_index[i * k + b - 1] > out_index[i * k + b]) {
                uint32_t t = out_index[i * k + b - 1];
                out_index[i * k + b - 1] = out_index[i * k + b];
                out_index[i * k + b] = t;
                b--;
            }
        }
    }
    if (ks.overflowed) *overflow_flag = 1ull;  // (any lane; the host repeats the call with a larger frontier)
}
