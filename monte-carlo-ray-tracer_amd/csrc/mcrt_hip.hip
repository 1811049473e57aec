// libmcrt_hip.so — host side of the gfx950 library: context, scene / photon-map upload, kernel selection and launches
// (megakernels and the wavefront frame loop), and the C ABI of include/mcrt.h. The kernels are in mcrt_kernels.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_integrator.hpp"
#include "mcrt_lanesm.hpp"
#include "mcrt_qbvh.hpp"
#include "mcrt_wavefront.hpp"
#include "mcrt_waveknn.hpp"
#include "mcrt_groupknn.hpp"
#include "mcrt_widerec.hpp"
#include "mcrt_layout.hpp"
#include "mcrt_internal.hpp"
#include "mcrt_plan.hpp"
#include "mcrt_octree_shared.hpp"
#include "mcrt_lean.hpp"

#include <hipcub/hipcub.hpp>

extern char** environ;

using namespace mcrt;

namespace {
#include "mcrt_kernels.hpp"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    // grow-only: keeps the allocation when it is already large enough (the operator-level entry points' scratch)
    hipError_t reserve(size_t n) { return n <= bytes ? hipSuccess : alloc(n); }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Work buffers that keep their allocation between calls: take() hands out the next buffer of the pool (the k-th take of a call
// gets the buffer the k-th take of the previous call got), rewind() starts a call.
struct DevPool {
    std::vector<std::unique_ptr<DevBuf>> bufs;
    size_t next = 0;
    void rewind() { next = 0; }
    DevBuf& take() {
        if (next == bufs.size()) bufs.emplace_back(new DevBuf());
        return *bufs[next++];
    }
};

}  // namespace

struct mcrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string error;
    int num_cus = 0;
    size_t max_lds = 0;        // dynamic LDS a kernel that shades may ask for
    size_t max_lds_trace = 0;  // ... a kernel that only walks the tree (no static LDS)

    bool has_scene = false;
    bool q_single = false;  // HostLayout::q_single of the uploaded scene
    std::vector<float> flat_pre_host;  // the flat loop's cull records, host copy: renderKernelFlatK takes them as a kernel argument
    DeviceScene scene{};
    DevBuf node_bounds, node_meta, nodes64, qblocks, quadrics, prim, flat_prim, flat_index, flat_pre, surf_v, surf_normal, surf_rec, surf_vn, surf_area, surf_material, surf_kind, materials,
        light_surface, light_cdf, sobol_tab;

    bool has_photons = false;
    PhotonMapView maps[2]{};
    DevBuf knn_spill;  // frontier spill lists of the wave-cooperative searches (mcrt_waveknn.hpp)
    DevBuf map_bounds[2], map_start[2], map_contained[2], map_next[2], map_leaf[2], map_photons[2], map_children[2], map_pos[2];
    const WideRec* map_children_ptr[2] = {nullptr, nullptr};
    uint32_t map_root_a[2] = {0, 0}, map_root_m[2] = {0, 0};
    uint32_t k_nearest = 50;
    int direct_visualization = 0;

    DevBuf samples;  // per-sample radiance of a pass of the chunked integrator kernels (RenderParams::samples)
    DevBuf work_counter, stats, spill, knn_res_d2, knn_res_idx, knn_visit_d2, knn_visit_oct, out_tmp;
    size_t spill_bytes = 0;
    uint32_t knn_lanes = 0, knn_k = 0;
    // per-lane photon search (the kernel of k > 768 and of MCRT_KERNEL=legacy): frontier entries per lane, grown - frame rendered again,
    // operator call repeated - when a search ran out (KnnScratch::max_visit); force_pm_lane: a photon-mapped frame whose wave-cooperative
    // searches overflowed THEIR frontier (128 register entries + a 1 024-entry list per wave) is rendered again by the per-lane kernel
    uint32_t knn_visit_cap = kMaxVisit, knn_visit_alloc = 0;
    bool force_pm_lane = false;

    // scratch of the operator-level entry points (mcrt_intersect / mcrt_knn / mcrt_sampler / mcrt_bsdf): kept between calls, grown
    // on demand, so that a host that only wants traversal or k-NN does not pay five hipMalloc / hipFree pairs per call
    DevBuf op_buf[6];
    std::map<std::string, std::string> options;  // mcrt_set_option; seeded from the MCRT_* environment variables at mcrt_create
    DevBuf pm_iors;  // refraction histories of the 1024-lane photon-mapping kernel
    // the frame in flight, kept so that mcrt_render_finish can run it again through the wavefront pipeline (deep refraction histories)
    mcrt_camera_desc last_cam;
    uint32_t last_seed = 0;
    int last_integrator = 0;
    double* last_out = nullptr;
    double* last_film = nullptr;
    hipStream_t last_stream = nullptr;
    bool force_wf = false;
    mutable bool lean_used = false;  // the last launch (frame, photon pass) ran a lean instance: mcrt_get_option("MCRT_LEAN_USED")
    uint32_t material_flags_or = 0xFFFFFFFFu;  // OR of the uploaded scene's material flags: which compiled-out features a lean kernel instance may lack (leanOf)
    uint32_t iors_depth = kMaxIorsDeep;  // RefractionHistory entries per pipeline slot (8 in LDS + deep rows); grows when a frame nests deeper
    DevBuf wf_iors_deep;
    DevPool pass_pool;  // work buffers of the device photon pass (mcrt_photon_device.hpp)
    DevBuf pm_stage; // estimate requests of the photon-mapping kernel, one record per resident lane (mcrt_waveknn.hpp)

    // photon emission pass
    std::vector<double> host_light_flux;  // [num_lights][3] emittance * area (photon-mapper.cpp:64)
    DevBuf emit_first, emit_flux, emit_counters, emit_photons[2], emit_keys[2];
    std::vector<float> host_photons[2];
    std::vector<uint64_t> host_keys[2];

    // wavefront path tracer: slot pool, ray queue, control words {count[2], pop}, pinned read-back word
    DevBuf wf_pool, wf_queue, wf_ctrl, wf_film, wf_film_cache, wf_requests, wf_res_n, wf_res_r2, wf_res_idx, wf_res_d2, wf_stage, wf_est;
    uint32_t wf_res_slots = 0, wf_res_k = 0;
    uint32_t wf_slots = 0;
    unsigned long long* wf_host = nullptr;   // pinned: one read-back word per half

    // in-flight render
    bool pending = false;
    std::chrono::steady_clock::time_point t_begin;
    uint32_t launches = 0;
    uint32_t kernel_id = MCRT_KERNEL_NONE;  // kernel form of the last render (mcrt_stats.kernel_id)
    bool dense_cell_refused = false;        // buildMapOnDevice: the 21-level cell codes could not separate a leaf's worth of photons
};

namespace {

int fail(mcrt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, MCRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

int planSampleStore(mcrt_ctx* ctx, uint32_t width, uint32_t owned_rows, uint32_t spp, PassPlan& pp);  // below

template <class T>
int uploadArray(mcrt_ctx* ctx, DevBuf& buf, const T* host, size_t count) {
    HIP_TRY(ctx, buf.alloc(count * sizeof(T)));
    if (count) HIP_TRY(ctx, hipMemcpy(buf.p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return MCRT_OK;
}


template <class T>
int uploadInto(mcrt_ctx* ctx, DevBuf& buf, const T* host, size_t count) {  // like uploadArray, into a grow-only buffer
    HIP_TRY(ctx, buf.reserve(std::max<size_t>(count * sizeof(T), 1)));
    if (count) HIP_TRY(ctx, hipMemcpy(buf.p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return MCRT_OK;
}

// Control words of the wavefront pipeline, one allocation of this size wherever it is made: {count[2], pop, -} per half of
// the pool (words 0..3 and 4..7; the photon mapper uses 4..6 as {rcount[2], rpop}). mcrt_intersect uses words 0..3.
constexpr size_t kWfCtrlWords = 8;

// Operator-level entry points and the emission pass share the context's stats buffer, events and scratch with a render:
// they are refused while one is in flight.
#define REJECT_IF_PENDING(ctx, what)                                                                                  \
    do {                                                                                                              \
        if ((ctx)->pending) return fail(ctx, MCRT_ERR_INVALID, what ": a render is in flight, call mcrt_render_finish first"); \
    } while (0)

struct LaunchGeom {
    uint32_t grid, lds_bytes, total_lanes, block = kBlock;
};

template <class K>
int launchGeometry(mcrt_ctx* ctx, K kernel, const DeviceScene& s, LaunchGeom& g, int plan = 0) {  // plan 2: flat-scene instance
    g.block = plan == 4 ? 1024u : plan == 3 ? 768u : kBlock;  // plan 2 / 3 / 4: the flat-scene instances (512 / 768 / 1024 lanes, no stack in LDS)
    g.lds_bytes = plan == 1 ? planSmLds(s, kBlock).total : planLds(s, g.block, plan < 2).total;
    if (g.lds_bytes > ctx->max_lds) return fail(ctx, MCRT_ERR_INVALID, "LDS plan exceeds the device limit");
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)g.lds_bytes));
    int per_cu = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)g.block, g.lds_bytes));
    if (per_cu < 1) per_cu = 1;
    g.grid = (uint32_t)(per_cu * ctx->num_cus);
    g.total_lanes = g.grid * g.block;
    return MCRT_OK;
}

int ensureSpill(mcrt_ctx* ctx, size_t bytes) {  // traversal-stack spill area, shared by every kernel (one render at a time)
    if (ctx->spill_bytes < bytes) {
        if (ctx->spill.alloc(bytes) != hipSuccess) {
            (void)hipGetLastError();
            ctx->spill_bytes = 0;
            return fail(ctx, MCRT_ERR_UNSUPPORTED, "the traversal stacks of this tree (" + std::to_string(ctx->scene.stack_depth) + " entries per ray: a depth-first walk may hold that many "
                                                   "pending nodes) need " + std::to_string(bytes >> 20) + " MiB of device memory, which could not be allocated");
        }
        ctx->spill_bytes = bytes;
    }
    return MCRT_OK;
}

int ensureScratch(mcrt_ctx* ctx, uint32_t total_lanes, bool photon) {
    if (!ctx->work_counter.p) HIP_TRY(ctx, ctx->work_counter.alloc(sizeof(unsigned long long)));
    if (!ctx->stats.p) HIP_TRY(ctx, ctx->stats.alloc(kStatsWords * sizeof(unsigned long long)));
    if (int rc = ensureSpill(ctx, (size_t)total_lanes * (ctx->scene.stack_depth - kLdsStackDepth) * sizeof(StackEntry))) return rc;
    if (photon && (ctx->knn_lanes < total_lanes || ctx->knn_k < ctx->k_nearest || ctx->knn_visit_alloc < ctx->knn_visit_cap)) {
        const uint32_t k = std::max<uint32_t>(ctx->k_nearest, 1);
        ctx->knn_lanes = ctx->knn_k = ctx->knn_visit_alloc = 0;
        HIP_TRY(ctx, ctx->knn_res_d2.alloc((size_t)total_lanes * k * sizeof(double)));
        HIP_TRY(ctx, ctx->knn_res_idx.alloc((size_t)total_lanes * k * sizeof(uint32_t)));
        HIP_TRY(ctx, ctx->knn_visit_d2.alloc((size_t)total_lanes * ctx->knn_visit_cap * sizeof(double)));
        HIP_TRY(ctx, ctx->knn_visit_oct.alloc((size_t)total_lanes * ctx->knn_visit_cap * sizeof(uint32_t)));
        ctx->knn_lanes = total_lanes;
        ctx->knn_k = k;
        ctx->knn_visit_alloc = ctx->knn_visit_cap;
    }
    return MCRT_OK;
}

PhotonMapViewW waveMapView(const mcrt_ctx* ctx, int which) {
    PhotonMapViewW v;
    v.base = ctx->maps[which];
    v.wide = ctx->map_children_ptr[which];
    v.root_a = ctx->map_root_a[which];
    v.root_m = ctx->map_root_m[which];
    v.pos = ctx->map_pos[which].as<PhotonPos>();
    return v;
}

// The positions of a map's photons by themselves (PhotonPos, mcrt_waveknn.hpp), from the records in map order.
__global__ void photonPosKernel(const float4* records, uint64_t n, PhotonPos* pos) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = records[2 * i], b = records[2 * i + 1];  // flux rgb, x | y, z, phi, theta
    pos[i] = PhotonPos{a.w, b.x, b.y};
}
int buildMapPositions(mcrt_ctx* ctx, int which, uint64_t n) {
    if (n == 0) return MCRT_OK;
    HIP_TRY(ctx, ctx->map_pos[which].reserve(n * sizeof(PhotonPos)));
    hipLaunchKernelGGL(photonPosKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->map_photons[which].as<float4>(), n,
                       ctx->map_pos[which].as<PhotonPos>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MCRT_OK;
}

int validateCamera(mcrt_ctx* ctx, const mcrt_camera_desc* cam) {
    if (!cam || cam->width == 0 || cam->height == 0 || cam->sqrtspp == 0)
        return fail(ctx, MCRT_ERR_INVALID, "camera: width, height and sqrtspp must be non-zero");
    if (cam->shard_count > 1 && cam->shard_index >= cam->shard_count)
        return fail(ctx, MCRT_ERR_INVALID, "camera: shard_index >= shard_count");
    if ((uint64_t)cam->width * cam->height > 0xFFFFFFFFull)
        return fail(ctx, MCRT_ERR_INVALID, "camera: more than 2^32 pixels");
    return MCRT_OK;
}

// Launch geometry of the trace kernel: one workgroup per CU (MCRT_TRACE_WAVES waves, default 16), its LDS split
// between the lanes' traversal stacks and as many top-of-tree child blocks as fit.
struct TracePlan {
    uint32_t grid, block, lds_bytes;
    WfTraceArgs args;
};

}  // namespace
namespace mcrt {
const char* ctxOpt(const mcrt_ctx* ctx, const char* key) {
    if (!ctx) return nullptr;
    auto it = ctx->options.find(key);
    return it == ctx->options.end() ? nullptr : it->second.c_str();
}
long ctxOptL(const mcrt_ctx* ctx, const char* key, long dflt) {
    const char* v = ctxOpt(ctx, key);
    return v ? atol(v) : dflt;
}
bool ctxOptOn(const mcrt_ctx* ctx, const char* key) {
    const char* v = ctxOpt(ctx, key);
    return v && atoi(v) != 0;
}
}  // namespace mcrt
namespace {
using mcrt::ctxOpt;
using mcrt::ctxOptL;
using mcrt::ctxOptOn;

// Lean kernel instances (csrc/mcrt_hip_lean.hip: the default path's kernels compiled without Oren-Nayar, GGX and conductor Fresnel). A
// scene whose materials carry none of those flags renders through them - same bits, fewer registers (mcrt_shade.hpp) - unless
// MCRT_LEAN_KERNELS=0. `full` is the instance the selection code above chose; the table says which of them has a lean twin.
bool leanScene(const mcrt_ctx* ctx) { return (ctx->material_flags_or & MCRT_LEAN_FEATURES_OFF) == 0u && ctxOptL(ctx, "MCRT_LEAN_KERNELS", 1) != 0; }
template <class K>
K leanOf(mcrt_ctx* ctx, K full) {
    if (!full || !leanScene(ctx)) return full;
    constexpr int PT = MCRT_INTEGRATOR_PATH_TRACER;
    static const struct { const void* full; int id; } twins[] = {
        {reinterpret_cast<const void*>(renderKernelFlatK<>), MCRT_LEAN_FLATK_512},
        {reinterpret_cast<const void*>(renderKernelFlatK<768>), MCRT_LEAN_FLATK_768},
        {reinterpret_cast<const void*>(renderKernel<PT, false, true, false, 1>), MCRT_LEAN_FLAT_512},
        {reinterpret_cast<const void*>(renderKernel<PT, false, true, false, 2>), MCRT_LEAN_FLAT_768},
        // (the photon-mapping kernel of trees in MEMORY keeps its full instance: lean it spills 883 registers instead of 769 and a C5 frame
        // takes 845 ms instead of 815 - that kernel's frame time follows its spill placement, not its instruction count, DESIGN 4.4 -
        // while the LDS-resident scenes' instance gains 7 %: profiles/r06_ab_lean_kernels.log)
        {reinterpret_cast<const void*>(renderKernelPM<false, true, 1024>), MCRT_LEAN_PM_1024_ALL},
        {reinterpret_cast<const void*>(renderKernelPM<false, true>), MCRT_LEAN_PM_512_ALL},
        {reinterpret_cast<const void*>(renderKernelSM<false, false>), MCRT_LEAN_SM},
        {reinterpret_cast<const void*>(renderKernelSM<false, true>), MCRT_LEAN_SM_ALL},
        {reinterpret_cast<const void*>(wfShadeKernel<false>), MCRT_LEAN_SHADE},
        {reinterpret_cast<const void*>(wfShadeKernel<true>), MCRT_LEAN_SHADE_PM},
        {reinterpret_cast<const void*>(emitKernel<false>), MCRT_LEAN_EMIT},
        {reinterpret_cast<const void*>(emitKernel<true>), MCRT_LEAN_EMIT_ALL},
        {reinterpret_cast<const void*>(wfKnnKernel<true>), MCRT_LEAN_KNN_EVAL},
    };
    const void* f = reinterpret_cast<const void*>(full);
    for (const auto& t : twins)
        if (t.full == f)
            if (const void* l = mcrt_lean_kernel(t.id)) {
                ctx->lean_used = true;
                return reinterpret_cast<K>(const_cast<void*>(l));
            }
    return full;
}

template <class K>
int planTrace(mcrt_ctx* ctx, K kernel, uint64_t max_items, TracePlan& tp) {
    auto envi = [ctx](const char* k, long d) { return ctxOptL(ctx, k, d); };
    const uint32_t waves = (uint32_t)std::min<long>(std::max<long>(envi("MCRT_TRACE_WAVES", 16), 1), kTraceMaxBlock / 64);
    tp.block = waves * 64u;
    const uint32_t lds_stack = (uint32_t)std::min<long>(std::max<long>(envi("MCRT_TRACE_STACK", kLdsStackDepth), 4), kLdsStackDepth);
    const uint32_t stack_bytes = lds_stack * tp.block * (uint32_t)sizeof(SmStackEntry);
    const long lds_cap = std::min<long>((long)ctx->max_lds_trace, envi("MCRT_TRACE_LDS", (long)ctx->max_lds_trace));
    if ((long)stack_bytes + 128 + (long)(waves * kShareMapBytes) > lds_cap) return fail(ctx, MCRT_ERR_INVALID, "trace kernel: traversal stacks exceed the LDS");
    const uint32_t lds_blocks = (uint32_t)std::min<uint64_t>(ctx->scene.num_qblocks, ((uint64_t)lds_cap - stack_bytes - 128u - waves * kShareMapBytes) / 64u);
    tp.lds_bytes = lds_blocks * 64u + stack_bytes + 64u + waves * kShareMapBytes + 64u;  // + the workgroup's queue cursor + the waves' shared-leaf maps + the root's record
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp.lds_bytes));
    int per_cu = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)tp.block, tp.lds_bytes));
    if (per_cu < 1) per_cu = 1;
    tp.grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * ctx->num_cus, (max_items + tp.block - 1) / tp.block);
    if (tp.grid < 1) tp.grid = 1;
    const uint32_t total_lanes = tp.grid * tp.block;
    if (!ctx->stats.p) HIP_TRY(ctx, ctx->stats.alloc(kStatsWords * sizeof(unsigned long long)));
    // a region holds stack_depth entries per lane whatever part of them lives in LDS
    if (int rc = ensureSpill(ctx, (size_t)total_lanes * ctx->scene.stack_depth * sizeof(StackEntry))) return rc;
    WfTraceArgs& ta = tp.args;
    memset(&ta, 0, sizeof(ta));
    ta.stats = ctx->stats.as<unsigned long long>();
    ta.nodes = ctx->scene.nodes64;
    ta.qblocks = ctx->scene.qblocks;
    ta.num_nodes = ctx->scene.q_nodes;
    ta.lds_blocks = lds_blocks;
    ta.q_root_a = ctx->scene.q_root_a;
    ta.q_root_m = ctx->scene.q_root_m;
    ta.prim = ctx->scene.prim;
    ta.spill = ctx->spill.as<SmStackEntry>();
    ta.total_lanes = total_lanes;
    ta.refill_lanes = (int)envi("MCRT_WF_REFILL", 16);  // (32 while the queue cursor was one global atomic)
    ta.leaf_lanes = (int)envi("MCRT_WF_LEAF", 16);  // (shared step, C3 64 spp: 8 / 12 / 16 / 20 pending lanes 412.7 / 402.1 / 398.1 / 402.3 ms; gating on 48-56 offered primitives instead: 398.4-399.0)
    ta.leaf_items = (int)envi("MCRT_WF_LEAF_ITEMS", 1 << 20);
    ta.min_inner = (int)envi("MCRT_WF_MININNER", 8);
    ta.lds_stack = (int)lds_stack;
    ta.max_stack = ctx->scene.stack_depth;
    ta.deal_shift = (uint32_t)std::min<long>(std::max<long>(envi("MCRT_WF_DEAL", 6), 6), 20);
    return MCRT_OK;
}

// The wavefront frame loop: shade(0), then trace(i), shade(i+1) until a shade launch queues no ray. The host
// looks at the queue length every few iterations (a launch with nothing to do costs microseconds), so the
// call returns when the frame is complete; mcrt_render_finish() then only collects the statistics.
// film_out != NULL (mcrt_render_film_device): the splats of this shard's samples stay in the caller's full-frame RGBW buffer
// and the resolve is left to mcrt_film_resolve_device, after the caller has summed the shards' buffers.
int launchWavefront(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, double* d_out, hipStream_t stream,
                    bool count_tests, bool photon, double* film_out = nullptr) {
    auto envi = [ctx](const char* k, long d) { return ctxOptL(ctx, k, d); };
    WfFrame fr;
    memset(&fr, 0, sizeof(fr));
    fr.cam = *cam;
    fr.global_seed = global_seed;
    fr.spp = cam->sqrtspp * cam->sqrtspp;
    const uint32_t owned_rows = mcrt_shard_rows(cam, nullptr);
    if (cam->width > 0xFFFFu || owned_rows > 0xFFFFu)  // kWfUnit keeps a slot's pixel as two 16-bit numbers
        return fail(ctx, MCRT_ERR_INVALID, "camera: the wavefront integrator takes at most 65535 columns and 65535 rows per shard");
    fr.tiles_x = (cam->width + 7) / 8;
    fr.film.type = MCRT_FILM_BOX;
    if (filmSplats(cam->film_filter, cam->film_radius)) {  // Film::Film(width, height, json), film.cpp:19-58
        if (cam->film_filter > MCRT_FILM_LANCZOS) return fail(ctx, MCRT_ERR_INVALID, "camera: unknown film filter");
        if (cam->shard_count > 1 && !film_out)
            return fail(ctx, MCRT_ERR_UNSUPPORTED, "reconstruction filters splat across row groups: render them unsharded (shard_count <= 1) "
                                                   "or with mcrt_render_film_device + mcrt_film_resolve_device");
        FilmView& f = fr.film;
        f.type = filmViewType(cam->film_filter);
        f.width = cam->width;
        f.height = cam->height;
        f.radius = cam->film_radius > 0.0 ? cam->film_radius : filmDefaultRadius(cam->film_filter);
        f.two_inv_radius = 2.0 / f.radius;
        f.cache_size = cam->film_cache_size;
        f.inv_dx = 0.0;
        f.cache = nullptr;
        if (f.cache_size) {
            if (f.cache_size < 2) return fail(ctx, MCRT_ERR_INVALID, "camera: film_cache_size must be 0 or at least 2");
            std::vector<double> table(f.cache_size);
            for (uint32_t i = 0; i < f.cache_size; i++) table[i] = filmFilterFunction(f.type, (2.0 * (int)i) / (double)(f.cache_size - 1));
            if (int rc = uploadArray(ctx, ctx->wf_film_cache, table.data(), table.size())) return rc;
            f.cache = ctx->wf_film_cache.as<double>();
            f.inv_dx = (double)(f.cache_size - 1) / f.radius;
        }
        const size_t blob_bytes = (size_t)cam->width * cam->height * 4 * sizeof(double);
        if (!film_out && ctx->wf_film.bytes < blob_bytes) HIP_TRY(ctx, ctx->wf_film.alloc(blob_bytes));
        f.blob = film_out ? film_out : ctx->wf_film.as<double>();
        HIP_TRY(ctx, hipMemsetAsync(f.blob, 0, blob_bytes, stream));
    }

    if (!ctx->work_counter.p) HIP_TRY(ctx, ctx->work_counter.alloc(sizeof(unsigned long long)));
    if (!ctx->stats.p) HIP_TRY(ctx, ctx->stats.alloc(kStatsWords * sizeof(unsigned long long)));
    ctx->t_begin = std::chrono::steady_clock::now();
    ctx->launches = 0;
    ctx->kernel_id = owned_rows == 0 ? MCRT_KERNEL_NONE : photon ? MCRT_KERNEL_WAVEFRONT_PM : MCRT_KERNEL_WAVEFRONT;
    HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter.p, 0, sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->stats.p, 0, kStatsWords * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    if (owned_rows == 0) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        ctx->pending = true;
        return MCRT_OK;
    }

    // Passes: as many rows as the per-sample store holds (box filter; splat frames keep no samples and are one pass).
    const bool splats = fr.film.type != MCRT_FILM_BOX;
    uint64_t pass_rows = owned_rows;
    if (!splats) {
        PassPlan pp;
        if (int rc = planSampleStore(ctx, cam->width, (uint32_t)owned_rows, fr.spp, pp)) return rc;
        pass_rows = pp.pass_rows;
        fr.samples = ctx->samples.as<double>();
    }
    // Pool size: up to 16 M slots (5.6 GB of pool, 4.6 GB of queue) — more slots = fewer, longer trace launches (their tails amortised; metal_bunnies
    // 1447 / 1492 / 1507 Mray/s with 4 / 8 / 16 M) — but no more than the pass has work for (below); units per pixel: the power of two
    // that gives a slot up to 16 work units of at least 4 samples.
    const uint64_t pixels = (uint64_t)cam->width * std::min<uint64_t>(pass_rows, owned_rows);
    // (round 4: 16 M by default - C3 at 1024 spp 2024 / 2068 / 2072 Mray/s with 8 / 16 / 32 M; 10 GB of pool and queue - but never more
    // than an eighth of the memory that is free on this device)
    uint64_t slots = (uint64_t)envi("MCRT_WF_SLOTS", 1l << 24);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t per_slot = (uint64_t)kWfWords * 8u + 2u * (2u * sizeof(uint32_t) + 2u * 8u * sizeof(double));
            const uint64_t have = (uint64_t)ctx->wf_slots * per_slot;  // what this context's pool and queue already hold
            slots = std::min<uint64_t>(slots, std::max<uint64_t>(((uint64_t)free_b + have) / 8u / per_slot, (uint64_t)kWfBlock));
        } else {
            (void)hipGetLastError();
        }
    }
    // ... and no more than the pass has work for: paths / 48, at least 2.5 M, never fewer than 4 samples per slot (planPoolSlots,
    // mcrt_plan.hpp: the measurements; options MCRT_WF_SLOT_PATHS, MCRT_WF_SLOT_FLOOR)
    // (photon-mapped frames: 16 path samples per slot - their iterations carry a kNN launch whose tails a larger pool amortises: C5 at
    // full size 2 757 ms with 48, 2 725 with 24, 2 713 with 12, 2 874 with 96: profiles/r06_ab_c5_pipeline_pool.log)
    slots = planPoolSlots(pixels * fr.spp, slots, kWfBlock, (uint64_t)std::max(1l, (long)envi("MCRT_WF_SLOT_PATHS", photon ? 16 : 48)),
                          (uint64_t)std::max(1l, (long)envi("MCRT_WF_SLOT_FLOOR", 2500000)), ctxOpt(ctx, "MCRT_WF_SLOT_PATHS") != nullptr);
    {
        const ChunkPlan cp = planChunks(fr.spp, unitsWanted(slots, 16, pixels, ctxOpt(ctx, "MCRT_CHUNKS")));
        fr.chunk_shift = cp.shift;
        fr.chunk = cp.chunk;
    }
    if (ctx->wf_slots != slots) {
        HIP_TRY(ctx, ctx->wf_pool.alloc((size_t)slots * kWfWords * 8));
        // ray queue, two entries per slot (bounce + shadow ray): item and light words, and two sets of eight planes of doubles
        // (WfRayQueue): a shade launch fills one set and reads the bounce rays of its slots back from the other
        HIP_TRY(ctx, ctx->wf_queue.alloc(((size_t)slots + 2 * kWfBlock) * 2 * (2 * sizeof(uint32_t) + 2 * 8 * sizeof(double))));
        ctx->wf_slots = (uint32_t)slots;
    }
    {   // deep refraction-history rows: [iors_depth - kMaxIors][slots] doubles (never initialised: an entry is written before it is read)
        const size_t need = (size_t)(ctx->iors_depth - kMaxIors) * slots * sizeof(double);
        if (ctx->wf_iors_deep.bytes < need) HIP_TRY(ctx, ctx->wf_iors_deep.alloc(need));
        fr.iors_deep = ctx->wf_iors_deep.as<double>();
        fr.iors_deep_rows = ctx->iors_depth - (uint32_t)kMaxIors;
    }
    if (!ctx->wf_ctrl.p) HIP_TRY(ctx, ctx->wf_ctrl.alloc(kWfCtrlWords * sizeof(unsigned long long)));
    if (!ctx->wf_host) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->wf_host), 2 * sizeof(unsigned long long)));
    auto runPass = [&]() -> int {  // the rows [fr.row_base, fr.row_end): shade / trace launches until nothing is queued
    HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter.p, 0, sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->wf_ctrl.p, 0, kWfCtrlWords * sizeof(unsigned long long), stream));
    // a fresh slot is all-zero flags (no path, no pixel); nothing else is used before it is written (kWfSeq is cleared too: the
    // sampler is rebuilt from it before the flags are looked at; tests/emu runs the same code on a pool of garbage)
    HIP_TRY(ctx, hipMemsetAsync(ctx->wf_pool.as<unsigned long long>() + (size_t)kWfFlags * slots, 0, (size_t)slots * 8, stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->wf_pool.as<unsigned long long>() + (size_t)kWfSeq * slots, 0, (size_t)slots * 8, stream));

    // MCRT_WF_LEAN (round 5; default 1): the inner visit is travInnerStepQLean - with one block per visit when the tree has no node with
    // more than four children (every quaternary tree); 0 (and MCRT_COUNT_TESTS): round 4's visit; 2: the block loop kept on a quaternary tree
    const int lean = !count_tests && ctxOptL(ctx, "MCRT_WF_LEAN", 1) != 0 ? (ctx->q_single && ctxOptL(ctx, "MCRT_WF_LEAN", 1) != 2 ? 3 : 1) : 0;
    void (*trace)(WfTraceArgs, PoolRays) = lean == 3   ? wfTraceKernel<PoolRays, false, 3>
                                           : lean == 1 ? wfTraceKernel<PoolRays, false, 1>
                                                       : (count_tests ? wfTraceKernel<PoolRays, true> : wfTraceKernel<PoolRays, false>);
    // + the materials and the light tables when they are small (MCRT_WF_LDS_TABLES=0: read them from memory)
    uint32_t shade_tables = wfShadeTableBytes(ctx->scene.num_materials, ctx->scene.num_lights);
    if (shade_tables > kWfShadeTableMax || ctxOptL(ctx, "MCRT_WF_LDS_TABLES", 1) == 0) shade_tables = 0;
    const uint32_t shade_lds = kSobolTableWords * 4u + kMaxIors * kWfBlock * 8u + shade_tables;
    // (the instances without the material features this scene does not use, when it uses none of them: leanOf)
    const auto shade_pt = leanOf(ctx, wfShadeKernel<false>);
    const auto shade_pm = leanOf(ctx, wfShadeKernel<true>);
    TracePlan tp;
    if (int rc = planTrace(ctx, trace, slots * 2, tp)) return rc;

    // control words: {count[2] (one per iteration parity), pop} of the ray queue; 4..6 = {rcount[2], rpop} of the estimate requests
    unsigned long long* ctrl = ctx->wf_ctrl.as<unsigned long long>();
    WfTraceArgs ta = tp.args;
    ta.pop = ctrl + 2;
    PoolRays pr;
    pr.pool.w = ctx->wf_pool.as<unsigned long long>();
    pr.pool.n = (uint32_t)slots;
    const size_t qcap = ((size_t)slots + 2 * kWfBlock) * 2;
    double* const ray_set0 = reinterpret_cast<double*>(ctx->wf_queue.as<uint32_t>() + 2 * qcap);  // iteration parity 0; parity 1: + 8 * qcap
    pr.q.item = ctx->wf_queue.as<uint32_t>();
    pr.q.light = pr.q.item + qcap;
    pr.q.ray = ray_set0;
    pr.q.prev_ray = ray_set0 + 8 * qcap;
    pr.q.cap = qcap;
    WfShadeArgs sa;
    memset(&sa, 0, sizeof(WfShadeArgs));
    sa.pool = pr.pool;
    sa.slot_base = 0u;
    sa.slot_count = (uint32_t)slots;
    sa.fr = fr;
    sa.queue = pr.q;
    sa.pop_reset = ctrl + 2;
    sa.work = ctx->work_counter.as<unsigned long long>();
    sa.stats = ctx->stats.as<unsigned long long>();
    sa.lds_tables = shade_tables;
    const uint32_t shade_grid = (sa.slot_count + kWfBlock - 1) / kWfBlock;

    // photon mapper: estimate requests and the kNN launch that serves them (control words 4..6 = {rcount[2], rpop})
    WfKnnArgs ka;
    memset(&ka, 0, sizeof(ka));
    uint32_t knn_grid = 0;
    bool knn_eval = false;
    if (photon) {
        const uint32_t k = ctx->k_nearest;
        // MCRT_WF_PM_EVAL (default 1): the kNN launch evaluates the estimates from staged Interactions; 0: it hands the k photons
        // back and the shade launch sums them per lane (round 2's form)
        knn_eval = ctxOptL(ctx, "MCRT_WF_PM_EVAL", 1) != 0;
        if (ctx->wf_res_slots != slots || ctx->wf_res_k != k || (knn_eval ? !ctx->wf_stage.p : !ctx->wf_res_idx.p)) {
            HIP_TRY(ctx, ctx->wf_requests.alloc((size_t)slots * sizeof(uint32_t)));
            if (knn_eval) {
                HIP_TRY(ctx, ctx->wf_stage.alloc((size_t)slots * kStageDoubles * sizeof(double)));
                HIP_TRY(ctx, ctx->wf_est.alloc((size_t)slots * 6 * sizeof(double)));
            } else {
                HIP_TRY(ctx, ctx->wf_res_n.alloc((size_t)2 * slots * sizeof(uint32_t)));
                HIP_TRY(ctx, ctx->wf_res_r2.alloc((size_t)2 * slots * sizeof(double)));
                HIP_TRY(ctx, ctx->wf_res_idx.alloc((size_t)2 * k * slots * sizeof(uint32_t)));
                HIP_TRY(ctx, ctx->wf_res_d2.alloc((size_t)2 * k * slots * sizeof(double)));
            }
            ctx->wf_res_slots = (uint32_t)slots;
            ctx->wf_res_k = k;
        }
        ka.pool = pr.pool;
        ka.requests = ctx->wf_requests.as<uint32_t>();
        ka.pop = ctrl + 6;
        ka.stats = ctx->stats.as<unsigned long long>();
        ka.maps[0] = waveMapView(ctx, 0);
        ka.maps[1] = waveMapView(ctx, 1);
        ka.k = k;
        ka.res_n = ctx->wf_res_n.as<uint32_t>();
        ka.res_r2 = ctx->wf_res_r2.as<double>();
        ka.res_idx = ctx->wf_res_idx.as<uint32_t>();
        ka.res_d2 = ctx->wf_res_d2.as<double>();
        ka.stage = knn_eval ? ctx->wf_stage.as<double>() : nullptr;
        ka.est = knn_eval ? ctx->wf_est.as<double>() : nullptr;
        knn_grid = (uint32_t)ctx->num_cus * 8u;
        HIP_TRY(ctx, ctx->knn_spill.reserve((size_t)knn_grid * 4 * kWaveSpill * 12));
        ka.spill = ctx->knn_spill.as<uint32_t>();
        WfShadeArgs& s0 = sa;
        s0.requests = ctx->wf_requests.as<uint32_t>();
        s0.rpop_reset = ctrl + 6;
        s0.pm.photons[0] = ctx->maps[0].photons;
        s0.pm.photons[1] = ctx->maps[1].photons;
        s0.pm.res_n = ka.res_n;
        s0.pm.res_r2 = ka.res_r2;
        s0.pm.res_idx = ka.res_idx;
        s0.pm.res_d2 = ka.res_d2;
        s0.pm.k = k;
        s0.pm.direct_visualization = ctx->direct_visualization != 0;
        s0.pm.est = ka.est;
        s0.stage = knn_eval ? ctx->wf_stage.as<double>() : nullptr;
    }

    // The host looks at the queue length every MCRT_WF_CHECK iterations (a launch with nothing to do costs microseconds)
    const uint64_t check_every = (uint64_t)std::max<long>(2, envi("MCRT_WF_CHECK", 16));
    for (uint64_t it = 0;; it++) {
        sa.count_out = ctrl + (it & 1);
        sa.count_reset = ctrl + ((it + 1) & 1);
        pr.q.ray = ray_set0 + (it & 1) * 8 * qcap;
        pr.q.prev_ray = ray_set0 + ((it + 1) & 1) * 8 * qcap;
        sa.queue = pr.q;
        if (photon) {
            sa.rcount_out = ctrl + 4 + (it & 1);
            sa.rcount_reset = ctrl + 4 + ((it + 1) & 1);
            hipLaunchKernelGGL(shade_pm, dim3(shade_grid), dim3(kWfBlock), shade_lds, stream, ctx->scene, sa);
        } else {
            hipLaunchKernelGGL(shade_pt, dim3(shade_grid), dim3(kWfBlock), shade_lds, stream, ctx->scene, sa);
        }
        ctx->launches++;
        if (it % check_every == check_every - 1 || it < 2) {
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(ctx->wf_host, ctrl + (it & 1), sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            if (photon)  // requests count as work too
                HIP_TRY(ctx, hipMemcpyAsync(ctx->wf_host + 1, ctrl + 4 + (it & 1), sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            if (ctxOptOn(ctx, "MCRT_WF_LOG"))  // queue length over the frame
                fprintf(stderr, "[mcrt wf] iteration %llu queued %llu at %.2f ms\n", (unsigned long long)it, ctx->wf_host[0],
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->t_begin).count());
            if (ctx->wf_host[0] == 0ull && (!photon || ctx->wf_host[1] == 0ull)) break;  // nothing queued: every slot is done
        }
        ta.count = ctrl + (it & 1);
        hipLaunchKernelGGL(trace, dim3(tp.grid), dim3(tp.block), tp.lds_bytes, stream, ta, pr);
        ctx->launches++;
        if (photon) {
            ka.count = ctrl + 4 + (it & 1);
            const bool large_k = ctx->k_nearest > waveMaxK(kWaveRows);  // the wide candidate buffer (mcrt_waveknn.hpp)
            if (knn_eval) hipLaunchKernelGGL((large_k ? wfKnnKernel<true, kWaveRowsLarge> : leanOf(ctx, wfKnnKernel<true>)), dim3(knn_grid), dim3(256), 0, stream, ka);
            else hipLaunchKernelGGL((large_k ? wfKnnKernel<false, kWaveRowsLarge> : wfKnnKernel<false>), dim3(knn_grid), dim3(256), 0, stream, ka);
            ctx->launches++;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return MCRT_OK;
    };
    for (uint32_t row = 0; row < owned_rows; row += (uint32_t)pass_rows) {
        fr.row_base = row;
        fr.row_end = (uint32_t)std::min<uint64_t>(owned_rows, row + pass_rows);
        fr.pass_pixels = (unsigned long long)(fr.row_end - fr.row_base) * cam->width;
        fr.work_items = ((unsigned long long)fr.tiles_x * ((fr.row_end - fr.row_base + 7) / 8) * 64ull) << fr.chunk_shift;
        if (int rc = runPass()) return rc;
        if (!splats) {  // the pass's samples, added up in sample order
            hipLaunchKernelGGL(sampleResolveKernel, dim3((uint32_t)((fr.pass_pixels + 255) / 256)), dim3(256), 0, stream, fr.samples,
                               (uint64_t)fr.pass_pixels, fr.spp, d_out + (size_t)fr.row_base * cam->width * 3);
            HIP_TRY(ctx, hipGetLastError());
            ctx->launches++;
        }
    }
    if (fr.film.type != MCRT_FILM_BOX && !film_out) {
        const uint64_t pixels = (uint64_t)cam->width * cam->height;
        hipLaunchKernelGGL(filmResolveKernel, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, stream, fr.film.blob, pixels, d_out);
        HIP_TRY(ctx, hipGetLastError());
        ctx->launches++;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
    ctx->pending = true;
    return MCRT_OK;
}

// Frames of DIFFERENT contexts on ONE device never overlap on the GPU: a launch waits (on the GPU, hipStreamWaitEvent) for the
// frame the device's previous launcher queued, and the host side of a launch — the whole frame for the wavefront pipeline — runs
// under the device's mutex. Two reasons. (i) The mutex makes "size this kernel's dynamic LDS, then launch it" one step:
// hipFuncSetAttribute is state of the kernel FUNCTION, shared by every context of the process - two contexts that render different
// scenes through the same kernel instance would otherwise race for it (a launch that asks for more LDS than the other context just
// set fails; loudly, but it fails). (ii) History: an experimental build of round 3 (hit records as one record per slot, never
// committed) gave wrong frames when two contexts of one process rendered on the same GPU at the same time, and was never understood.
// The committed kernels do not show it: tools/shared_gpu_stress.py with the ordering switched off - 3 processes x 2 contexts and 2 x 4,
// every kernel form, dirty memory, 1 680 concurrent frames compared bit for bit with the reference's / the oracle's - found none wrong
// (round 5, profiles/r05_shared_gpu_stress.log). So the GPU-side wait is a belt; the mutex is needed. One context per device, the
// production shape, never waits here.
struct DeviceOrder {
    std::mutex m;
    hipEvent_t last = nullptr;
    mcrt_ctx* owner = nullptr;
};
DeviceOrder g_device_order[64];

int launchRenderImpl(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_out, hipStream_t stream,
                     double* film_out);
int launchRender(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_out, hipStream_t stream,
                 double* film_out = nullptr) {
    if (!ctx) return MCRT_ERR_INVALID;
    // Option MCRT_DEVICE_ORDER=0 (per context, like every option; TEST ONLY: tools/shared_gpu_stress.py, which looks for the fault the
    // ordering was added against): this context's frames do not wait on the GPU for the device's previous launcher. The mutex stays
    // either way - "size this kernel's dynamic LDS, then launch it" must be one step (DeviceOrder's comment).
    const bool gpu_wait = ctxOptL(ctx, "MCRT_DEVICE_ORDER", 1) != 0;
    DeviceOrder& o = g_device_order[(unsigned)ctx->device & 63u];
    std::lock_guard<std::mutex> guard(o.m);
    if (gpu_wait && o.owner && o.owner != ctx) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamWaitEvent(stream, o.last, 0));
    }
    const int rc = launchRenderImpl(ctx, cam, global_seed, integrator, d_out, stream, film_out);
    if (rc == MCRT_OK && ctx->pending) {
        o.last = ctx->ev1;
        o.owner = ctx;
        if (cam != &ctx->last_cam) ctx->last_cam = *cam;
        ctx->last_seed = global_seed;
        ctx->last_integrator = integrator;
        ctx->last_out = d_out;
        ctx->last_film = film_out;
        ctx->last_stream = stream;
    }
    return rc;
}

int launchRenderImpl(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_out, hipStream_t stream,
                     double* film_out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, MCRT_ERR_NO_SCENE, "mcrt_render before mcrt_upload_scene");
    if (int rc = validateCamera(ctx, cam)) return rc;
    const bool photon = integrator == MCRT_INTEGRATOR_PHOTON_MAPPER;
    if (integrator != MCRT_INTEGRATOR_PATH_TRACER && !photon) return fail(ctx, MCRT_ERR_INVALID, "unknown integrator");
    if (photon && !ctx->has_photons) return fail(ctx, MCRT_ERR_NO_PHOTONS, "photon mapping render before mcrt_upload_photons");
    if (ctx->pending) return fail(ctx, MCRT_ERR_INVALID, "a render is already in flight: call mcrt_render_finish");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->lean_used = false;

    const bool count_tests = ctxOptOn(ctx, "MCRT_COUNT_TESTS");
    using KernelT = void (*)(const DeviceScene, const RenderParams);
    const bool all = ctx->scene.stage_all != 0;
    constexpr int PT = MCRT_INTEGRATOR_PATH_TRACER, PM = MCRT_INTEGRATOR_PHOTON_MAPPER;
    static const KernelT table[2][2][2] = {
        {{renderKernel<PT, false, false>, renderKernel<PT, false, true>}, {renderKernel<PT, true, false>, renderKernel<PT, true, true>}},
        {{renderKernel<PM, false, false>, renderKernel<PM, false, true>}, {renderKernel<PM, true, false>, renderKernel<PM, true, true>}}};
    KernelT kernel = table[photon ? 1 : 0][count_tests ? 1 : 0][all ? 1 : 0];
    const bool profile_phases = ctxOptOn(ctx, "MCRT_PROFILE_PHASES");
    if (profile_phases && !photon) kernel = all ? renderKernel<PT, false, true, true> : renderKernel<PT, false, false, true>;
    const bool flat_only = !photon && ctx->scene.flat && ctx->scene.flat_pre && all && !count_tests && !profile_phases && !ctxOptOn(ctx, "MCRT_FLAT_GENERIC");
    // Flat-mode scenes get their own instance of the kernel: without the BVH walk in the code it needs no traversal stack
    // (64 KB of LDS at 512 lanes), so a CU can hold more waves; measured on the C2 frame (ms): generic instance 862,
    // flat instance with 512 lanes (2 waves/SIMD, 256 VGPRs, no scratch) 795, 768 lanes (3, 168 VGPRs, 336 B/lane of
    // scratch) 716, 1024 lanes (4, 128 VGPRs, 524 B/lane) 690 — the FP64 dependency chains of the primitive tests and the
    // BSDF code want the extra waves more than they mind the spills.
    // With the FP32 cull in front of the FP64 tests (mcrt_scene.hpp) the order is reversed: 512 lanes 448.7 ms, 768 lanes 455.8,
    // 1024 lanes 485.2 (split next-event estimate: 454 / 493 / 563) — fewer instructions per ray, and the spills of the
    // narrow instances (107 / 169 VGPRs) now cost more than the extra waves hide.
    // MCRT_FLAT_KARG (round 5, default 1): the cull records travel in the kernel's argument block and are read with scalar loads
    // (renderKernelFlatK) - when they fit it. With the records in SGPRs the 768-lane shape (3 waves per SIMD, 168 VGPRs) is the
    // fastest: C2 439.6 ms against 442.8 at 512 lanes and 478 at 1024, C2-GGX 596.9 against 614.6 and 649
    // (profiles/r05_ab_c2_flat_karg.log) - so that is the default where the argument-block form applies, 512 lanes elsewhere.
    const bool flat_karg = flat_only && ctxOptL(ctx, "MCRT_FLAT_KARG", 1) != 0 && !ctx->flat_pre_host.empty() && ctx->flat_pre_host.size() <= kFlatPreArgFloats &&
                           ctx->flat_pre_host.size() == (size_t)ctx->scene.pre_tri_pairs * kTriPairFloats + (size_t)ctx->scene.pre_sph_pairs * kSphPairFloats;
    // (... of the full instance. The lean one - a scene without rough / conductor materials, leanOf - spills NOTHING at 512 lanes and is
    // fastest there: C2 108.2 ms per 64-spp frame against 111.0 at 768 lanes and the full instance's 112.0, profiles/r06_ab_feature_strip_probe.log)
    int flat_block = (int)ctxOptL(ctx, "MCRT_FLAT_BLOCK", flat_karg && !leanScene(ctx) ? 768 : 512);
    if (flat_block != 512 && flat_block != 768 && flat_block != 1024) flat_block = 512;
    // (Round 4 built a form that dealt a wave's (ray, cull survivor) pairs over all 64 lanes for the FP64 tests; measured in round 5 it
    // LOST 8 % on C2 and on C2-GGX - 482 ms against 446, 663 against 614, profiles/r05_ab_c2_flat_share.log - and was removed.)
    if (flat_only)
        kernel = leanOf(ctx, flat_block == 1024 ? renderKernel<PT, false, true, false, 3>
                                                : flat_block == 768 ? renderKernel<PT, false, true, false, 2> : renderKernel<PT, false, true, false, 1>);
    // path tracing of scenes whose BVH is walked: lane-state-machine kernel (MCRT_KERNEL=legacy keeps the
    // wave-synchronous one for A/B runs)
    const char* kenv = ctxOpt(ctx, "MCRT_KERNEL");
    const bool use_sm = !photon && !ctx->scene.flat && !(kenv && strcmp(kenv, "legacy") == 0);
    // ... and when the tree lives in HBM, the wavefront pipeline (MCRT_KERNEL=sm keeps the megakernel, MCRT_KERNEL=wf
    // forces the wavefront pipeline for any scene that has a BVH)
    const bool filtered = filmSplats(cam->film_filter, cam->film_radius);  // per-sample splats: the wavefront pipeline's shade kernel has them
    if (film_out && !filtered)
        return fail(ctx, MCRT_ERR_INVALID, "mcrt_render_film_device is for splatted frames (a reconstruction filter, or the box filter with a radius other than 0.5)");
    // (a scene without a BVH is walked through a tree over index ranges by the pipeline's trace kernel, mcrt_layout.hpp)
    const bool has_tree = ctx->scene.q_nodes > 0;
    if (filtered && !has_tree)
        return fail(ctx, MCRT_ERR_UNSUPPORTED, "reconstruction filters need the wavefront pipeline, and this scene has neither a BVH nor finite surface bounds to build its stand-in from");
    // Film::Film(w, h, json) with "filter": "box" and a radius other than the default 0.5 splats too (film.cpp:27-30); that case
    // is not built, so it is refused rather than rendered as the default box
    constexpr uint32_t kWaveKMax = waveMaxK(kWaveRowsLarge);  // 768: what the wave-cooperative search's widest candidate buffer serves
    if (filtered && photon && ctx->k_nearest > kWaveKMax)
        return fail(ctx, MCRT_ERR_UNSUPPORTED, "reconstruction filters on photon-mapped frames need k_nearest_photons <= 768 (wavefront pipeline)");
    const bool want_wf = filtered || (kenv && strcmp(kenv, "wf") == 0) || ctx->force_wf;
    // measured: the pipeline wins on deep trees (metal_bunnies 169 k nodes +28 %, spaceship with hulls 154 k
    // nodes +7 %), the megakernel on small ones (spaceship cockpit 23 k nodes: 1352 vs 940 Mray/s)
    const char* mn = ctxOpt(ctx, "MCRT_WF_MIN_NODES");
    const uint32_t wf_min_nodes = mn ? (uint32_t)strtoul(mn, nullptr, 0) : 65536u;
    // ... and, since round 4's trace kernel, on ANY tree in memory once the frame is large enough to amortise the pipeline's launches
    // (spaceship cockpit, 23 k nodes, 1080p, ms per frame megakernel / pipeline: 2 M paths 9.4 / 19.1, 8 M 24.5 / 35.3, 33 M 80.9 /
    // 77.3, 133 M 311 / 228): MCRT_WF_MIN_PATHS path samples in this call's rows, default 32 M
    const char* mp = ctxOpt(ctx, "MCRT_WF_MIN_PATHS");
    const uint64_t wf_min_paths = mp ? strtoull(mp, nullptr, 0) : 32000000ull;
    const uint64_t frame_paths = (uint64_t)mcrt_shard_rows(cam, nullptr) * cam->width * cam->sqrtspp * cam->sqrtspp;
    if (!photon && has_tree && (want_wf || (use_sm && !all && !kenv && (ctx->scene.num_nodes >= wf_min_nodes || frame_paths >= wf_min_paths))))
        return launchWavefront(ctx, cam, global_seed, d_out, stream, count_tests, false, film_out);
    // photon-mapped frames go through the pipeline (trace / kNN / shade launches) on request only: measured slower than
    // renderKernelPM (C5 9.3 vs 7.4 s per frame, hexagon_room map 308 vs 242 ms) — the kNN search is bound by the number of
    // wave instructions per query (one query per wave leaves most lanes idle), which more waves per SIMD do not fix, and
    // the pipeline adds its shade launches on top. k must fit the per-wave candidate buffer.
    // Round 6: ... and by itself for a scene whose tree stays in memory, whose materials allow the lean instances (leanOf) and whose k
    // fits the narrow buffers, once the frame is large enough (MCRT_WF_PM_MIN_PATHS path samples in this call's rows, default 32 M). With
    // the lean kNN launch (17 instead of 61 spilled registers, 6 waves per SIMD) and the lean shade launch (8 instead of 192) C5 renders
    // in 739 ms per 64-spp frame against the megakernel's 817 (profiles/r06_ab_lean_knn_occupancy.log): the pipeline's kernels each run at
    // their own register budget, the megakernel's estimates at the budget of its bounce code. LDS-resident scenes stay with the
    // megakernel (hexagon_room_pm 93.7 ms against 136).
    const char* pmp = ctxOpt(ctx, "MCRT_WF_PM_MIN_PATHS");
    const uint64_t wf_pm_min_paths = pmp ? strtoull(pmp, nullptr, 0) : 32000000ull;
    const bool pm_pipeline = photon && has_tree && !all && !kenv && !count_tests && leanScene(ctx) && ctx->k_nearest <= waveMaxK(kWaveRows) &&
                             frame_paths >= wf_pm_min_paths;
    if (photon && has_tree && ctx->k_nearest <= kWaveKMax && (want_wf || pm_pipeline) && !ctx->force_pm_lane)
        return launchWavefront(ctx, cam, global_seed, d_out, stream, count_tests, true, film_out);
    // workgroup size of the state-machine kernel for trees that stay in HBM (MCRT_SM_BLOCK: 512 / 768 / 1024 lanes) and the
    // stack entries per lane it keeps in LDS (MCRT_SM_STACK; the rest of a lane's stack is in the HBM spill area)
    int sm_block = (int)kBlock, sm_depth = kLdsStackDepth;
    if (use_sm) {
        static const KernelT sm_table[2][2] = {{renderKernelSM<false, false>, renderKernelSM<false, true>},
                                               {renderKernelSM<true, false>, renderKernelSM<true, true>}};
        kernel = leanOf(ctx, sm_table[count_tests ? 1 : 0][all ? 1 : 0]);
        if (profile_phases) kernel = all ? renderKernelSM<false, true, true> : renderKernelSM<false, false, true>;
        const int want = (int)ctxOptL(ctx, "MCRT_SM_BLOCK", (long)kBlock);
        if (!all && !count_tests && !profile_phases && (want == 768 || want == 1024)) {
            sm_block = want;
            sm_depth = want == 768 ? 8 : 6;
            kernel = want == 768 ? renderKernelSM<false, false, false, 768> : renderKernelSM<false, false, false, 1024>;
        }
        if (ctxOpt(ctx, "MCRT_SM_STACK")) sm_depth = std::min(std::max((int)ctxOptL(ctx, "MCRT_SM_STACK", 0), 2), (int)kLdsStackDepth);
    }

    // photon mapping: wave-cooperative estimates unless k is too large for the widest per-wave buffer (k <= 128: 256 candidates per
    // wave; k <= 768: 1024 candidates per wave, 512 lanes per workgroup)
    bool use_pm_wave = photon && ctx->k_nearest <= kWaveKMax && !(kenv && strcmp(kenv, "legacy") == 0) && !ctx->force_pm_lane;
    const bool pm_large_k = use_pm_wave && ctx->k_nearest > waveMaxK(kWaveRows);
    if (pm_large_k) {
        // the wide buffers take 100 KB of a 512-lane workgroup's LDS: a BVH staged whole with its 16 stack entries per lane may not
        // leave that (a tree in HBM keeps as few as 2 entries per lane in LDS, a flat scene has no stack) - then the per-lane kernel
        DeviceScene probe = ctx->scene;
        if (!probe.stage_all) probe.stage_nodes = std::min<uint32_t>(probe.stage_nodes, 128u);
        const uint32_t least = alignUp(planLds(probe, kBlock, true, probe.stage_all ? (uint32_t)kLdsStackDepth : 2u, kPmLdsIors).total, 16) +
                               (kBlock / 64) * (waveKnnBytes(kWaveRowsLarge) + kWaveStateBytes);
        if (least > ctx->max_lds) use_pm_wave = false;
    }
    using PmKernelT = void (*)(const DeviceScene, const RenderParams, const PmExtra);
    PmKernelT pm_kernel = nullptr;
    using FlatKT = void (*)(const DeviceScene, const RenderParams, const FlatPreArg);
    FlatKT flatk_kernel = nullptr;
    DeviceScene launch_scene = ctx->scene;
    if (ctxOpt(ctx, "MCRT_FLAT_CULL") && !ctxOptOn(ctx, "MCRT_FLAT_CULL")) launch_scene.flat_pre = nullptr;  // A/B: every primitive in FP64
    LaunchGeom g;
    uint32_t pm_stack_depth = kLdsStackDepth;
    if (use_pm_wave) {
        static const PmKernelT pm_table[2][2][2] = {{{renderKernelPM<false, false>, renderKernelPM<false, true>},
                                                     {renderKernelPM<true, false>, renderKernelPM<true, true>}},
                                                    {{renderKernelPM<false, false, 1024>, renderKernelPM<false, true, 1024>},
                                                     {renderKernelPM<true, false, 1024>, renderKernelPM<true, true, 1024>}}};
        if (!launch_scene.stage_all) launch_scene.stage_nodes = std::min<uint32_t>(launch_scene.stage_nodes, 128u);
        // 1024 lanes per workgroup (4 waves per SIMD) when the LDS plan allows it: flat scenes have no traversal stack; a tree in
        // HBM is walked with the state machine's stack, of which then only a few entries per lane stay in LDS (the rest
        // spills to HBM); a staged BVH walked by the wave-synchronous code needs its 16 entries (512 lanes).
        static const PmKernelT pm_table_large[2][2] = {{renderKernelPM<false, false, (int)kBlock, kWaveRowsLarge>, renderKernelPM<false, true, (int)kBlock, kWaveRowsLarge>},
                                                       {renderKernelPM<true, false, (int)kBlock, kWaveRowsLarge>, renderKernelPM<true, true, (int)kBlock, kWaveRowsLarge>}};
        const int want = pm_large_k ? (int)kBlock : (int)ctxOptL(ctx, "MCRT_PM_BLOCK", 1024);
        const uint32_t knn_bytes = waveKnnBytes(pm_large_k ? kWaveRowsLarge : kWaveRows) + (all ? 0u : kWaveStateBytes);
        g.block = kBlock;
        // (the 1024-lane instance keeps two refraction-history entries per lane in LDS, the deeper ones in global memory)
        auto ldsBytes = [&](uint32_t block, uint32_t depth) {
            return alignUp(planLds(launch_scene, block, true, depth, (block != 1024u && !pm_large_k) ? (uint32_t)kMaxIors : kPmLdsIors).total, 16) + (block / 64) * knn_bytes;
        };
        if (pm_large_k && !launch_scene.stage_all) {
            pm_stack_depth = 2;
            for (uint32_t depth = 16u; depth > 2u; depth -= 2)
                if (ldsBytes(kBlock, depth) <= ctx->max_lds) {
                    pm_stack_depth = depth;
                    break;
                }
        }
        // (round 5: a 768-lane instance - 3 waves per SIMD, 168 VGPRs, 639 instead of 769 spill instructions - measured 895 ms against
        // 762 on the C5 probe and 112 against 101 on pm, profiles/r05_ab_pm768.log: this kernel wants its four waves)
        if (want == 1024) {
            const uint32_t wb = 1024u;
            if (launch_scene.flat && ldsBytes(wb, kLdsStackDepth) <= ctx->max_lds) {
                g.block = wb;
            } else if (!launch_scene.stage_all) {
                const uint32_t depth_max = ctxOpt(ctx, "MCRT_PM_STACK") ? (uint32_t)std::max(2, (int)ctxOptL(ctx, "MCRT_PM_STACK", 16)) & ~1u : 16u;
                for (uint32_t depth = depth_max; depth >= 2 && g.block == kBlock; depth -= 2)
                    if (ldsBytes(wb, depth) <= ctx->max_lds) {
                        g.block = wb;
                        pm_stack_depth = depth;
                    }
            }
        }
        pm_kernel = pm_large_k ? pm_table_large[count_tests ? 1 : 0][all ? 1 : 0] : pm_table[g.block == 1024 ? 1 : 0][count_tests ? 1 : 0][all ? 1 : 0];
        pm_kernel = leanOf(ctx, pm_kernel);
        g.lds_bytes = ldsBytes(g.block, pm_stack_depth);
        if (g.lds_bytes > ctx->max_lds) return fail(ctx, MCRT_ERR_INVALID, "LDS plan exceeds the device limit");
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(pm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes));
        int per_cu = 0;
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pm_kernel, (int)g.block, g.lds_bytes));
        if (per_cu < 1) per_cu = 1;
        g.grid = (uint32_t)(per_cu * ctx->num_cus);
        g.total_lanes = g.grid * g.block;
    } else if (use_sm) {
        // the staged top of the tree shrinks to what the larger workgroup's stacks and refraction histories leave
        g.block = (uint32_t)sm_block;
        const uint32_t fixed = planSmLds(DeviceScene{}, g.block, (uint32_t)sm_depth).total;
        if (!launch_scene.stage_all && fixed < ctx->max_lds)
            launch_scene.stage_nodes = std::min<uint32_t>(launch_scene.stage_nodes, (ctx->max_lds - fixed) / 64u);
        g.lds_bytes = planSmLds(launch_scene, g.block, (uint32_t)sm_depth).total;
        if (g.lds_bytes > ctx->max_lds) return fail(ctx, MCRT_ERR_INVALID, "LDS plan exceeds the device limit");
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes));
        int per_cu = 0;
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)g.block, g.lds_bytes));
        if (per_cu < 1) per_cu = 1;
        g.grid = (uint32_t)(per_cu * ctx->num_cus);
        g.total_lanes = g.grid * g.block;
    } else if (flat_karg && launch_scene.flat_pre) {
        flatk_kernel = leanOf(ctx, flat_block == 1024 ? renderKernelFlatK<1024> : flat_block == 768 ? renderKernelFlatK<768> : renderKernelFlatK<>);
        if (int rc = launchGeometry(ctx, flatk_kernel, launch_scene, g, flat_block == 1024 ? 4 : flat_block == 768 ? 3 : 2)) return rc;
    } else if (int rc = launchGeometry(ctx, kernel, launch_scene, g, flat_only ? (flat_block == 1024 ? 4 : flat_block == 768 ? 3 : 2) : 0)) {
        return rc;
    }
    if (int rc = ensureScratch(ctx, g.total_lanes, photon && !use_pm_wave)) return rc;
    if (use_sm && sm_depth < kLdsStackDepth)
        if (int rc = ensureSpill(ctx, (size_t)g.total_lanes * (ctx->scene.stack_depth - sm_depth) * sizeof(StackEntry))) return rc;
    if (use_pm_wave && pm_stack_depth < (uint32_t)kLdsStackDepth)
        if (int rc = ensureSpill(ctx, (size_t)g.total_lanes * (ctx->scene.stack_depth - pm_stack_depth) * sizeof(StackEntry))) return rc;

    RenderParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.cam = *cam;
    prm.global_seed = global_seed;
    prm.spp = cam->sqrtspp * cam->sqrtspp;
    prm.owned_rows = mcrt_shard_rows(cam, nullptr);
    prm.tiles_x = (cam->width + 7) / 8;
    prm.tiles_y = (prm.owned_rows + 7) / 8;
    prm.work_items = (uint64_t)prm.tiles_x * prm.tiles_y * 64ull;
    prm.work_counter = ctx->work_counter.as<unsigned long long>();
    prm.stats = ctx->stats.as<unsigned long long>();
    prm.spill = ctx->spill.as<StackEntry>();
    prm.total_lanes = g.total_lanes;
    {
        auto envi = [ctx](const char* k, int d) { return (int)ctxOptL(ctx, k, d); };
        prm.sm_shade_lanes = envi("MCRT_SM_SHADE", 40);
        prm.sm_regen_lanes = envi("MCRT_SM_REGEN", 16);
        prm.sm_min_trav = envi("MCRT_SM_MINTRAV", 20);
        prm.sm_leaf_lanes = envi("MCRT_SM_LEAF", 32);
        prm.sm_min_inner = envi("MCRT_SM_MININNER", 8);
        prm.sm_lds_depth = sm_depth;
    }
    if (photon) {
        prm.global_map = ctx->maps[0];
        prm.caustic_map = ctx->maps[1];
        prm.k_nearest = ctx->k_nearest;
        prm.direct_visualization = (uint32_t)ctx->direct_visualization;
        prm.knn_res_d2 = ctx->knn_res_d2.as<double>();
        prm.knn_res_idx = ctx->knn_res_idx.as<uint32_t>();
        prm.knn_visit_d2 = ctx->knn_visit_d2.as<double>();
        prm.knn_visit_oct = ctx->knn_visit_oct.as<uint32_t>();
        prm.knn_max_visit = ctx->knn_visit_alloc;
    }
    ctx->kernel_id = prm.owned_rows == 0 ? MCRT_KERNEL_NONE
                     : use_pm_wave   ? MCRT_KERNEL_PM_WAVE
                     : photon        ? MCRT_KERNEL_PM_LANE
                     : use_sm        ? MCRT_KERNEL_LANE_SM
                     : flat_only     ? MCRT_KERNEL_FLAT
                                     : MCRT_KERNEL_WAVESYNC;
    if (prm.owned_rows == 0) {
        ctx->pending = true;
        ctx->launches = 0;
        ctx->t_begin = std::chrono::steady_clock::now();
        HIP_TRY(ctx, hipMemsetAsync(ctx->stats.p, 0, kStatsWords * sizeof(unsigned long long), stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
        return MCRT_OK;
    }
    ctx->t_begin = std::chrono::steady_clock::now();
    ctx->launches = 0;
    HIP_TRY(ctx, hipMemsetAsync(ctx->stats.p, 0, kStatsWords * sizeof(unsigned long long), stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    {
        // Sample-chunked work units (RenderParams): the frame goes through in passes of as many rows as the per-sample
        // store holds (MCRT_SAMPLE_STORE_GB, default 64: mcrt_plan.hpp), each pass = one
        // integrator launch + the in-order resolve.
        PassPlan pp;
        if (int rc = planSampleStore(ctx, cam->width, prm.owned_rows, prm.spp, pp)) return rc;
        const uint64_t pass_rows = pp.pass_rows;
        prm.samples = ctx->samples.as<double>();
        PmExtra pmx;
        if (use_pm_wave) {
            pmx.global_map = waveMapView(ctx, 0);
            pmx.caustic_map = waveMapView(ctx, 1);
            pmx.stack_depth = pm_stack_depth;
            pmx.iors_global = nullptr;
            HIP_TRY(ctx, ctx->pm_stage.reserve((size_t)kStageDoubles * g.total_lanes * sizeof(double)));
            pmx.stage = ctx->pm_stage.as<double>();
            HIP_TRY(ctx, ctx->knn_spill.reserve((size_t)(g.total_lanes / 64) * kWaveSpill * 12));
            pmx.knn_spill = ctx->knn_spill.as<uint32_t>();
            if (g.block == 1024u || pm_large_k) {  // (two refraction-history entries per lane in LDS, the deeper ones in memory)
                HIP_TRY(ctx, ctx->pm_iors.reserve((size_t)kMaxIors * g.total_lanes * sizeof(double)));
                pmx.iors_global = ctx->pm_iors.as<double>();
            }
        }
        for (uint32_t row = 0; row < prm.owned_rows; row += (uint32_t)pass_rows) {
            prm.row_base = row;
            prm.row_end = (uint32_t)std::min<uint64_t>(prm.owned_rows, row + pass_rows);
            prm.pass_pixels = (uint64_t)(prm.row_end - prm.row_base) * cam->width;
            // units per pixel: a power of two that gives every resident lane >= 128 units in chunks of at least 16 samples
            // (planChunksMega, mcrt_plan.hpp: the measurements behind it)
            // (photon-mapped frames keep the short chunks: their paths differ far more in cost - a search per diffuse hit - and the
            // balance is worth more than the units' fixed cost: C5 at 64 spp 770 ms with 64 units of 4 samples, 791 with 16 of 16)
            const ChunkPlan cp = photon ? planChunks(prm.spp, unitsWanted(g.total_lanes, 128, prm.pass_pixels, ctxOpt(ctx, "MCRT_CHUNKS")))
                                        : planChunksMega(prm.spp, g.total_lanes, prm.pass_pixels, ctxOpt(ctx, "MCRT_CHUNKS"));
            const uint32_t shift = cp.shift;
            prm.chunk_shift = cp.shift;
            prm.chunk = cp.chunk;
            const uint64_t tiles = (uint64_t)prm.tiles_x * ((prm.row_end - prm.row_base + 7) / 8);
            prm.work_items = (tiles * 64ull) << shift;
            // never launch more lanes than there is work
            const uint32_t grid = (uint32_t)std::min<uint64_t>(g.grid, (prm.work_items + g.block - 1) / g.block);
            HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter.p, 0, sizeof(unsigned long long), stream));
            if (use_pm_wave) {
                hipLaunchKernelGGL(pm_kernel, dim3(grid), dim3(g.block), g.lds_bytes, stream, launch_scene, prm, pmx);
            } else if (flat_karg && launch_scene.flat_pre) {
                FlatPreArg pre;
                memset(&pre, 0, sizeof(pre));
                memcpy(pre.v, ctx->flat_pre_host.data(), ctx->flat_pre_host.size() * sizeof(float));
                hipLaunchKernelGGL(flatk_kernel, dim3(grid), dim3(g.block), g.lds_bytes, stream, launch_scene, prm, pre);
            } else {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(g.block), g.lds_bytes, stream, launch_scene, prm);
            }
            hipLaunchKernelGGL(sampleResolveKernel, dim3((uint32_t)((prm.pass_pixels + 255) / 256)), dim3(256), 0, stream, prm.samples,
                               prm.pass_pixels, prm.spp, d_out + (size_t)prm.row_base * cam->width * 3);
            ctx->launches += 2;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
    ctx->pending = true;
    return MCRT_OK;
}


// The per-sample store of a frame (mcrt_plan.hpp): MCRT_SAMPLE_STORE_GB (default 64) is an upper bound, the device decides how much of
// it exists - at most 40 % of the memory that is free now plus what this context's store already holds (hipMemGetInfo), so that a
// 1080p @ 1024 spp frame (51 GB in one pass on a 288 GB MI355X) goes through in more passes on a smaller or shared device instead of
// failing with out-of-memory; if the allocation still fails (fragmentation, another process grew meanwhile) the store is halved
// until it fits or one 8-row pass does not. Returns the plan through `pp`.
int planSampleStore(mcrt_ctx* ctx, uint32_t width, uint32_t owned_rows, uint32_t spp, PassPlan& pp) {
    double gb = sampleStoreGb(ctxOpt(ctx, "MCRT_SAMPLE_STORE_GB"));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) gb = std::min(gb, 0.4 * (double)(free_b + ctx->samples.bytes) / 1e9);
    else (void)hipGetLastError();
    for (;;) {
        pp = planPasses(width, owned_rows, spp, gb);
        if (ctx->samples.bytes >= pp.store_bytes) return MCRT_OK;
        if (ctx->samples.alloc(pp.store_bytes) == hipSuccess) return MCRT_OK;
        (void)hipGetLastError();
        if (pp.pass_rows <= 8) return fail(ctx, MCRT_ERR_HIP, "out of device memory for the per-sample store of one 8-row pass");
        gb = std::min(gb, (double)pp.store_bytes / 1e9) * 0.5;
    }
}

int uploadMap(mcrt_ctx* ctx, int which, const mcrt_photon_map_desc* m) {
    PhotonMapView& v = ctx->maps[which];
    memset(&v, 0, sizeof(v));
    ctx->map_children_ptr[which] = nullptr;
    if (!m || m->num_octants == 0 || m->num_photons == 0) return MCRT_OK;
    if (m->num_photons > 0xFFFFFFFEull) return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon map larger than 2^32-2 photons per GPU");
    if (!m->octant_bounds || !m->octant_start_data || !m->octant_contained_data || !m->octant_next_sibling || !m->octant_leaf || !m->photons)
        return fail(ctx, MCRT_ERR_INVALID, "photon map descriptor has null arrays");
    const size_t n = m->num_octants;
    std::vector<uint32_t> start(n), contained(n);
    for (size_t i = 0; i < n; i++) {
        if (m->octant_start_data[i] + m->octant_contained_data[i] > m->num_photons)
            return fail(ctx, MCRT_ERR_INVALID, "photon map octant range exceeds the photon array");
        start[i] = (uint32_t)m->octant_start_data[i];
        contained[i] = (uint32_t)m->octant_contained_data[i];
    }
    if (int rc = uploadArray(ctx, ctx->map_bounds[which], m->octant_bounds, n * 6)) return rc;
    if (int rc = uploadArray(ctx, ctx->map_start[which], start.data(), n)) return rc;
    if (int rc = uploadArray(ctx, ctx->map_contained[which], contained.data(), n)) return rc;
    if (int rc = uploadArray(ctx, ctx->map_next[which], m->octant_next_sibling, n)) return rc;
    if (int rc = uploadArray(ctx, ctx->map_leaf[which], m->octant_leaf, n)) return rc;
    if (int rc = uploadArray(ctx, ctx->map_photons[which], m->photons, (size_t)m->num_photons * 8)) return rc;
    {   // record lists for the wave-cooperative search (mcrt_waveknn.hpp: WideRec; built by buildWideRecords, mcrt_widerec.hpp)
        std::vector<WideRec> wide;
        uint32_t root_a = 0, root_m = 0;
        const int wrc = buildWideRecords(m, contained.data(), std::max<uint32_t>(ctx->k_nearest, 1u), wide, root_a, root_m);
        if (wrc == 1) return fail(ctx, MCRT_ERR_INVALID, "photon octant with more than 8 children");
        if (wrc == 2) return fail(ctx, MCRT_ERR_UNSUPPORTED, "photon map too large for 32-bit record indices");
        if (int rc = uploadArray(ctx, ctx->map_children[which], wide.data(), wide.size())) return rc;
        ctx->map_children_ptr[which] = ctx->map_children[which].as<WideRec>();
        ctx->map_root_a[which] = root_a;
        ctx->map_root_m[which] = root_m;
    }
    v.num_octants = m->num_octants;
    v.num_photons = m->num_photons;
    v.octant_bounds = ctx->map_bounds[which].as<double>();
    v.octant_start = ctx->map_start[which].as<uint32_t>();
    v.octant_contained = ctx->map_contained[which].as<uint32_t>();
    v.octant_next = ctx->map_next[which].as<uint32_t>();
    v.octant_leaf = ctx->map_leaf[which].as<uint8_t>();
    v.photons = ctx->map_photons[which].as<float>();
    return buildMapPositions(ctx, which, m->num_photons);
}

#include "mcrt_photon_device.hpp"

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int mcrt_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count > 0 ? count : 0;
}

int mcrt_create(mcrt_ctx** out, int device_id) {
    if (!out) return MCRT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, MCRT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(nullptr, MCRT_ERR_NO_DEVICE, "device_id out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return fail(nullptr, MCRT_ERR_NO_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0 && !getenv("MCRT_ALLOW_ANY_ARCH"))
        return fail(nullptr, MCRT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if ((e = hipSetDevice(device_id)) != hipSuccess)
        return fail(nullptr, MCRT_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    mcrt_ctx* ctx = new mcrt_ctx();
    // The MCRT_* environment variables seed the context's options HERE, once; afterwards only mcrt_set_option changes them
    // (no getenv on the launch path).
    for (char** e = environ; e && *e; e++) {
        if (strncmp(*e, "MCRT_", 5) != 0) continue;
        const char* eq = strchr(*e, '=');
        if (eq) ctx->options[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->max_lds = prop.sharedMemPerBlock > 0 ? (size_t)prop.sharedMemPerBlock : 65536;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && v > 0)
            ctx->max_lds = std::max(ctx->max_lds, (size_t)v);
    }
    // the kernels that shade hold the sin/cos table as static LDS (mcrt_libm.hpp): their dynamic LDS plans get the rest
    ctx->max_lds_trace = ctx->max_lds;
    ctx->max_lds -= glibc235::kShadeStaticLds;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return fail(nullptr, MCRT_ERR_HIP, "stream/event creation failed");
    }
    std::vector<uint32_t> tab(kSobolTableWords);
    buildSobolByteTables(tab.data());
    if (int rc = uploadArray(ctx, ctx->sobol_tab, tab.data(), tab.size())) {
        g_create_error = ctx->error;
        mcrt_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MCRT_OK;
}

void mcrt_destroy(mcrt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    {
        DeviceOrder& o = g_device_order[(unsigned)ctx->device & 63u];
        std::lock_guard<std::mutex> guard(o.m);
        if (o.owner == ctx) {  // nobody may wait on an event that is about to go
            (void)hipEventSynchronize(o.last);
            o.owner = nullptr;
            o.last = nullptr;
        }
    }
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->wf_host) (void)hipHostFree(ctx->wf_host);
    delete ctx;
}

const char* mcrt_last_error(const mcrt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mcrt_set_option(mcrt_ctx* ctx, const char* key, const char* value) {
    if (!ctx || !key || strncmp(key, "MCRT_", 5) != 0) return ctx ? fail(ctx, MCRT_ERR_INVALID, "mcrt_set_option: keys are the MCRT_* names of include/mcrt.h") : MCRT_ERR_INVALID;
    if (value) ctx->options[key] = value;
    else ctx->options.erase(key);
    return MCRT_OK;
}

const char* mcrt_get_option(const mcrt_ctx* ctx, const char* key) {
    if (!ctx || !key) return nullptr;
    if (strcmp(key, "MCRT_LEAN_USED") == 0) return ctx->lean_used ? "1" : "0";  // (read-only: did the last frame / photon pass run a lean kernel instance)
    return mcrt::ctxOpt(ctx, key);
}

int mcrt_upload_scene(mcrt_ctx* ctx, const mcrt_scene_desc* s) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!s || s->abi_version != MCRT_ABI_VERSION) return fail(ctx, MCRT_ERR_INVALID, "scene descriptor: wrong abi_version");
    REJECT_IF_PENDING(ctx, "mcrt_upload_scene");
    if (s->num_surfaces == 0 || !s->surf_kind || !s->surf_interpolate || !s->surf_material || !s->surf_area || !s->surf_v || !s->surf_e ||
        !s->materials || s->num_materials == 0)
        return fail(ctx, MCRT_ERR_INVALID, "scene descriptor: missing surface/material arrays");
    if (s->num_nodes && (!s->node_bounds || !s->node_start_surface || !s->node_num_surfaces || !s->node_next_sibling))
        return fail(ctx, MCRT_ERR_INVALID, "scene descriptor: missing node arrays");
    if (s->num_lights && (!s->light_surface || !s->light_cdf)) return fail(ctx, MCRT_ERR_INVALID, "scene descriptor: missing light arrays");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->has_scene = false;

    const size_t ns = s->num_surfaces;
    HostLayout L;
    std::string lerr;
    if (int rc = buildLayout(s, L, lerr)) return fail(ctx, rc, lerr);
    const bool any_vn = L.any_vn;
    std::vector<double>& prim = L.prim;
    std::vector<double>& normal = L.normal;
    std::vector<double>& bounds = L.node_bounds;
    std::vector<NodeMeta>& meta = L.node_meta;

    if (int rc = uploadArray(ctx, ctx->node_bounds, bounds.data(), bounds.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->node_meta, meta.data(), meta.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->nodes64, L.nodes64.data(), L.nodes64.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->qblocks, L.qblocks.data(), L.qblocks.size())) return rc;
    // quadric records first: the primitive records and surf_v of quadric surfaces carry their device addresses
    if (L.num_quadric_surfaces) {
        for (uint32_t i = 0; i < s->num_lights; i++)
            if (s->surf_kind[s->light_surface[i]] == MCRT_SURF_QUADRIC)
                return fail(ctx, MCRT_ERR_UNSUPPORTED, "emissive quadrics are not supported (scene/scene.cpp:125)");
        if (int rc = uploadArray(ctx, ctx->quadrics, s->quadrics, (size_t)s->num_quadrics * 22)) return rc;
        patchQuadricAddresses(s, L, ctx->quadrics.as<double>());
    } else {
        ctx->quadrics.release();
    }
    if (int rc = uploadArray(ctx, ctx->prim, prim.data(), prim.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->flat_prim, L.flat_prim.data(), L.flat_prim.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->flat_index, L.flat_index.data(), L.flat_index.size())) return rc;
    ctx->flat_pre_host = L.flat_pre;
    if (L.flat_pre.empty()) ctx->flat_pre.release();
    else if (int rc = uploadArray(ctx, ctx->flat_pre, L.flat_pre.data(), L.flat_pre.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->surf_v, L.num_quadric_surfaces ? L.surf_v_patched.data() : s->surf_v, ns * 9)) return rc;
    if (int rc = uploadArray(ctx, ctx->surf_normal, normal.data(), normal.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->surf_rec, L.shade_rec.data(), L.shade_rec.size())) return rc;
    if (any_vn) {
        if (int rc = uploadArray(ctx, ctx->surf_vn, s->surf_vn, ns * 9)) return rc;
    } else {
        ctx->surf_vn.release();
    }
    if (int rc = uploadArray(ctx, ctx->surf_area, s->surf_area, ns)) return rc;
    if (int rc = uploadArray(ctx, ctx->surf_material, s->surf_material, ns)) return rc;
    if (int rc = uploadArray(ctx, ctx->surf_kind, s->surf_kind, ns)) return rc;
    if (int rc = uploadArray(ctx, ctx->materials, s->materials, (size_t)s->num_materials)) return rc;
    ctx->material_flags_or = 0u;
    for (uint32_t i = 0; i < s->num_materials; i++) ctx->material_flags_or |= s->materials[i].flags;
    if (int rc = uploadArray(ctx, ctx->light_surface, s->light_surface, (size_t)s->num_lights)) return rc;
    if (int rc = uploadArray(ctx, ctx->light_cdf, s->light_cdf, (size_t)s->num_lights)) return rc;

    DeviceScene& d = ctx->scene;
    memset(&d, 0, sizeof(d));
    d.num_nodes = s->num_nodes;
    d.num_surfaces = s->num_surfaces;
    d.num_materials = s->num_materials;
    d.num_lights = s->num_lights;
    d.node_bounds = ctx->node_bounds.as<double>();
    d.node_meta = ctx->node_meta.as<NodeMeta>();
    d.nodes64 = ctx->nodes64.as<Node64>();
    d.qblocks = ctx->qblocks.as<QBlock>();
    d.num_qblocks = (uint32_t)L.qblocks.size();
    d.q_nodes = (uint32_t)L.nodes64.size();
    // every lane's traversal stack holds what a depth-first walk of THIS tree can hold (HostLayout::stack_bound), never fewer than
    // kMaxStackDepth entries; the spill slabs behind the LDS part are sized from it at launch (ensureSpill)
    d.stack_depth = std::max<uint32_t>((uint32_t)kMaxStackDepth, L.stack_bound + 1u);
    {
        // the deepest spill slab a frame allocates: two trace launches x 256 CUs x the 2048 lanes a CU can hold (planTrace's grid is
        // occupancy x CUs) x 8 bytes per entry. A tree degenerate enough to need more than a quarter of the device's memory for it
        // (tens of thousands of stack entries: a BVH that is a list) is refused here, with its number; ensureSpill says the same
        // should an allocation below that bound fail all the same.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            total_b = (size_t)64 << 30;
        }
        const size_t slab = (size_t)d.stack_depth * sizeof(StackEntry) * 2u * (size_t)ctx->num_cus * 2048u;
        if (slab > total_b / 4)
            return fail(ctx, MCRT_ERR_UNSUPPORTED, "BVH so unbalanced that a depth-first walk may hold " + std::to_string(L.stack_bound) +
                                                       " pending nodes per ray: the traversal stacks would not fit in device memory");
    }
    d.q_root_a = L.q_root_a;
    d.q_root_m = L.q_root_m;
    ctx->q_single = L.q_single;
    d.prim = ctx->prim.as<double>();
    d.flat_prim = ctx->flat_prim.as<double>();
    d.flat_index = ctx->flat_index.as<uint32_t>();
    d.flat_tris = L.flat_tris;
    d.flat_pre = L.flat_pre.empty() ? nullptr : ctx->flat_pre.as<float>();
    d.pre_tri_pairs = L.pre_tri_pairs;
    d.pre_sph_pairs = L.pre_sph_pairs;
    for (int c = 0; c < 3; c++) d.pre_centre[c] = L.pre_centre[c];
    d.pre_bound = L.pre_bound;
    d.surf_v = ctx->surf_v.as<double>();
    d.surf_normal = ctx->surf_normal.as<double>();
    d.surf_rec = ctx->surf_rec.as<double>();
    d.surf_vn = any_vn ? ctx->surf_vn.as<double>() : nullptr;
    d.surf_area = ctx->surf_area.as<double>();
    d.surf_material = ctx->surf_material.as<uint32_t>();
    d.surf_kind = ctx->surf_kind.as<uint8_t>();
    d.materials = ctx->materials.as<mcrt_material>();
    d.light_surface = ctx->light_surface.as<uint32_t>();
    d.light_cdf = ctx->light_cdf.as<double>();
    d.sobol_tab = ctx->sobol_tab.as<uint32_t>();
    d.scene_ior = s->scene_ior;

    ctx->host_light_flux.assign((size_t)s->num_lights * 3, 0.0);
    for (uint32_t i = 0; i < s->num_lights; i++) {
        const uint32_t ls = s->light_surface[i];
        for (int c = 0; c < 3; c++) ctx->host_light_flux[(size_t)i * 3 + c] = s->materials[s->surf_material[ls]].emittance[c] * s->surf_area[ls];
    }

    // Staging plan: whole scene when its LDS image is <= 48 KiB, else the top 512 nodes of the BVH.
    d.stage_all = 1;
    d.stage_nodes = 0;
    const uint32_t fixed = planLds(DeviceScene{}, kBlock).total;
    // (... and only when the plan of the 512-lane kernels - tables, stacks, histories AND the image - fits the device's LDS: an image of
    // 40-48 KiB did not, and its mcrt_intersect / legacy frames failed with "LDS plan exceeds the device limit" until round 4)
    if (planLds(d, kBlock).total - fixed > 48u * 1024u || planLds(d, kBlock).total > ctx->max_lds ||
        L.num_quadric_surfaces) {  // quadric code lives in the kAll == false kernels
        d.stage_all = 0;
        d.stage_nodes = std::min<uint32_t>(d.num_nodes, 512u);
    }
    // Tiny scenes: a BVH of a few dozen primitives costs more in wavefront divergence (every lane walks
    // its own node sequence) than it saves in tests. With <= MCRT_FLAT_MAX primitives (default 64) all
    // lanes test all primitives in one wave-uniform loop, as Scene::intersect does without a "bvh" key
    // (scene.cpp:161-173); the closest hit is the same.
    const char* fm = ctxOpt(ctx, "MCRT_FLAT_MAX");
    const uint32_t flat_max = fm ? (uint32_t)strtoul(fm, nullptr, 0) : 64u;
    d.flat = (d.stage_all && d.num_surfaces <= flat_max && !L.flat_prim.empty() && L.num_quadric_surfaces == 0) ? 1u : 0u;  // the flat loop knows triangles and spheres
    ctx->has_scene = true;
    return MCRT_OK;
}

int mcrt_upload_photons(mcrt_ctx* ctx, const mcrt_photon_map_desc* global_map, const mcrt_photon_map_desc* caustic_map,
                        uint32_t k_nearest_photons, int direct_visualization) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (k_nearest_photons == 0) return fail(ctx, MCRT_ERR_INVALID, "k_nearest_photons must be > 0");
    REJECT_IF_PENDING(ctx, "mcrt_upload_photons");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->has_photons = false;
    ctx->k_nearest = k_nearest_photons;  // before the maps: the search's record lists expand octants with more than k photons
    if (int rc = uploadMap(ctx, 0, global_map)) return rc;
    if (int rc = uploadMap(ctx, 1, caustic_map)) return rc;
    ctx->direct_visualization = direct_visualization ? 1 : 0;
    ctx->has_photons = true;
    return MCRT_OK;
}

uint32_t mcrt_shard_rows(const mcrt_camera_desc* cam, uint32_t* rows) {
    if (!cam) return 0;
    if (cam->shard_count <= 1) {
        if (rows)
            for (uint32_t y = 0; y < cam->height; y++) rows[y] = y;
        return cam->height;
    }
    const uint32_t g = cam->shard_rows ? cam->shard_rows : 1;
    uint32_t n = 0;
    for (uint32_t y = 0; y < cam->height; y++)
        if ((y / g) % cam->shard_count == cam->shard_index) {
            if (rows) rows[n] = y;
            n++;
        }
    return n;
}

int mcrt_render_device(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_out_rgb,
                       void* stream) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!d_out_rgb) return fail(ctx, MCRT_ERR_INVALID, "d_out_rgb is NULL");
    return launchRender(ctx, cam, global_seed, integrator, d_out_rgb, stream ? (hipStream_t)stream : ctx->stream);
}

int mcrt_render_film_device(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* d_rgbw,
                            void* stream) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!d_rgbw) return fail(ctx, MCRT_ERR_INVALID, "d_rgbw is NULL");
    return launchRender(ctx, cam, global_seed, integrator, nullptr, stream ? (hipStream_t)stream : ctx->stream, d_rgbw);
}

int mcrt_film_resolve_device(mcrt_ctx* ctx, uint32_t width, uint32_t height, const double* d_rgbw, double* d_out_rgb, void* stream) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!d_rgbw || !d_out_rgb || width == 0 || height == 0) return fail(ctx, MCRT_ERR_INVALID, "mcrt_film_resolve_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t pixels = (uint64_t)width * height;
    hipLaunchKernelGGL(filmResolveKernel, dim3((uint32_t)((pixels + 255) / 256)), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream,
                       d_rgbw, pixels, d_out_rgb);
    HIP_TRY(ctx, hipGetLastError());
    return MCRT_OK;
}

int mcrt_render_finish(mcrt_ctx* ctx, mcrt_stats* stats) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!ctx->pending) return fail(ctx, MCRT_ERR_INVALID, "no render in flight");
    ctx->pending = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    unsigned long long h[kStatsWords];
    HIP_TRY(ctx, hipMemcpy(h, ctx->stats.p, sizeof(h), hipMemcpyDeviceToHost));
    if (ctxOptOn(ctx, "MCRT_PROFILE_PHASES")) {
        static const char* names[kNumPhases] = {"regen", "trav/inner", "shade", "shadow/leaf", "sample", "loop"};
        unsigned long long tw = 0;
        for (int i = 0; i < kNumPhases; i++) tw += h[8 + i];
        for (int i = 0; i < kNumPhases; i++)
            fprintf(stderr, "[mcrt phase] %-9s wave-cycles %6.2f%%  lane utilisation %5.1f%%\n", names[i], 100.0 * h[8 + i] / (double)(tw ? tw : 1),
                    h[8 + i] ? 100.0 * h[8 + kNumPhases + i] / (64.0 * h[8 + i]) : 0.0);
    }
    if (ctx->kernel_id == MCRT_KERNEL_WAVEFRONT && h[8] && ctxOptOn(ctx, "MCRT_COUNT_TESTS"))
        fprintf(stderr, "[mcrt trace] per wave iteration: %.1f lanes hold a ray; inner step in %.1f%% of the iterations with %.1f lanes, leaf step in %.1f%% with %.1f lanes, "
                        "%.1f leaf lanes wait; wave cycles: inner %.1f%%, leaf %.1f%%, rest %.1f%% (of the kernel: refills %.1f%%, pop site %.1f%%); per ray: %.2f inner steps, %.2f leaf steps\n",
                (double)h[9] / h[8], 100.0 * h[10] / h[8], h[10] ? (double)h[11] / h[10] : 0.0, 100.0 * h[12] / h[8], h[12] ? (double)h[13] / h[12] : 0.0,
                (double)h[14] / h[8], 100.0 * h[15] / (double)h[17], 100.0 * h[16] / (double)h[17], 100.0 * (h[17] - h[15] - h[16]) / (double)h[17],
                100.0 * h[18] / (double)h[17], 100.0 * h[19] / (double)h[17],
                (double)h[11] / (double)(h[1] ? h[1] : 1), (double)h[13] / (double)(h[1] ? h[1] : 1));
    if (ctx->kernel_id == MCRT_KERNEL_PM_WAVE && h[9] && ctxOptOn(ctx, "MCRT_COUNT_TESTS"))
        fprintf(stderr, "[mcrt pm] wave cycles inside the radiance estimates: %.1f%% of the kernel (%llu searches, %.1f octants per search)\n",
                100.0 * (double)h[8] / (double)h[9], h[4], h[4] ? (double)h[6] / (double)h[4] : 0.0);
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->paths = h[0];
        stats->rays = h[1];
        stats->node_tests = h[2];
        stats->prim_tests = h[3];
        stats->knn_searches = h[4];
        stats->kernel_ms = ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->t_begin).count();
        stats->kernel_launches = ctx->launches;
        stats->kernel_id = ctx->kernel_id;
    }
    // (test hook: MCRT_TEST_KNN_OVERFLOW=1 treats the first frame of a wave-cooperative photon-mapping kernel as if a search had overflowed;
    // no tree the tests can build in reasonable time fills 128 + 1 024 frontier entries through the render path, whose record lists
    // are made for the k it searches with)
    const bool pm_wave_frame = ctx->kernel_id == MCRT_KERNEL_PM_WAVE || ctx->kernel_id == MCRT_KERNEL_WAVEFRONT_PM;
    if (pm_wave_frame && !ctx->force_pm_lane && ctxOptOn(ctx, "MCRT_TEST_KNN_OVERFLOW")) h[5] |= kKnnOverflowFlag;
    if (h[5] >= kKnnOverflowFlag) {
        // The reference's frontier is an unbounded priority queue (linear-octree.cpp:33). A wave-cooperative search keeps 128 entries
        // in registers and 1 024 in a list in memory; a frame in which one of them ran out is rendered AGAIN by the per-lane kernel (the
        // reference's two queues per lane, in memory), whose own frontier - 160 entries per lane to begin with - grows eightfold per
        // attempt, up to kMaxVisitLimit. Slower, and correct (round 6; until then: MCRT_ERR_UNSUPPORTED, and the per-lane search DROPPED
        // the entry without a word).
        const bool lane_frame = ctx->kernel_id == MCRT_KERNEL_PM_LANE;
        const bool splats = filmSplats(ctx->last_cam.film_filter, ctx->last_cam.film_radius);  // (only the pipeline splats: no second kernel for such a frame)
        if (!splats && ((pm_wave_frame && !ctx->force_pm_lane) || (lane_frame && ctx->knn_visit_cap < kMaxVisitLimit))) {
            if (lane_frame) ctx->knn_visit_cap = std::min<uint32_t>(ctx->knn_visit_cap * 8u, kMaxVisitLimit);
            else ctx->knn_visit_cap = std::max<uint32_t>(ctx->knn_visit_cap, 2048u);
            const bool keep = ctx->force_pm_lane;
            ctx->force_pm_lane = true;
            const int rc = launchRender(ctx, &ctx->last_cam, ctx->last_seed, ctx->last_integrator, ctx->last_out, ctx->last_stream, ctx->last_film);
            if (rc != MCRT_OK) {
                ctx->force_pm_lane = keep;
                return rc;
            }
            const int rc2 = mcrt_render_finish(ctx, stats);
            ctx->force_pm_lane = keep;
            return rc2;
        }
        return fail(ctx, MCRT_ERR_UNSUPPORTED, "kNN frontier overflow: a search had more than " + std::to_string(kMaxVisitLimit) + " octants pending at once in the per-lane "
                                               "kernel's frontier (the reference's queue is unbounded, linear-octree.cpp:33)");
    }
    if (h[5])
        return fail(ctx, MCRT_ERR_UNSUPPORTED, "traversal stack overflow (internal error: the stacks are sized to the tree's own bound, HostLayout::stack_bound)");
    if (h[7]) {
        // RefractionHistory (ray.cpp:74-98) is an unbounded vector. The megakernels keep kMaxIors (8) entries per lane, the pipeline
        // ctx->iors_depth per slot (32 to begin with). A frame that nested deeper is rendered AGAIN: a megakernel frame through the
        // pipeline, a pipeline frame with four times the rows (round 6; until then the frame failed beyond 32) - slower, and correct.
        // The rows a scene needed stay with the context. (No scene of the reference nests deeper than 4.)
        const bool was_pipeline = ctx->kernel_id == MCRT_KERNEL_WAVEFRONT || ctx->kernel_id == MCRT_KERNEL_WAVEFRONT_PM;
        const bool photon = ctx->last_integrator == MCRT_INTEGRATOR_PHOTON_MAPPER;
        const bool can_pipeline = ctx->scene.q_nodes > 0 && (!photon || ctx->k_nearest <= waveMaxK(kWaveRowsLarge));
        constexpr uint32_t kIorsDepthLimit = 1u << 15;  // (32 768 nested media - a slot keeps the history's size in 16 bits; beyond it something other than a scene is going on)
        if (can_pipeline && ((!was_pipeline && !ctx->force_wf) || (was_pipeline && ctx->iors_depth < kIorsDepthLimit))) {
            if (was_pipeline) ctx->iors_depth *= 4u;
            const bool keep = ctx->force_wf;
            ctx->force_wf = true;
            const int rc = launchRender(ctx, &ctx->last_cam, ctx->last_seed, ctx->last_integrator, ctx->last_out, ctx->last_stream, ctx->last_film);
            ctx->force_wf = keep;
            if (rc != MCRT_OK) return rc;
            return mcrt_render_finish(ctx, stats);
        }
        return fail(ctx, MCRT_ERR_UNSUPPORTED, "a path entered more nested dielectric media than this frame can keep (RefractionHistory, ray.cpp:74-98: 8 per lane in the "
                                               "megakernels of scenes the pipeline cannot take; 32 768 per slot in the pipeline)");
    }
    return MCRT_OK;
}

int mcrt_render(mcrt_ctx* ctx, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, double* out_rgb,
                mcrt_stats* stats) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!out_rgb) return fail(ctx, MCRT_ERR_INVALID, "out_rgb is NULL");
    if (int rc = validateCamera(ctx, cam)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t rows = mcrt_shard_rows(cam, nullptr);
    const size_t row_bytes = (size_t)cam->width * 3 * sizeof(double);
    if (ctx->out_tmp.bytes < rows * row_bytes) HIP_TRY(ctx, ctx->out_tmp.alloc(std::max<size_t>(rows * row_bytes, 8)));
    if (int rc = launchRender(ctx, cam, global_seed, integrator, ctx->out_tmp.as<double>(), ctx->stream)) return rc;
    mcrt_stats st;
    int rc = mcrt_render_finish(ctx, &st);
    if (rc) return rc;
    if (rows) {
        std::vector<double> packed((size_t)rows * cam->width * 3);
        HIP_TRY(ctx, hipMemcpy(packed.data(), ctx->out_tmp.p, packed.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<uint32_t> idx(rows);
        mcrt_shard_rows(cam, idx.data());
        for (uint32_t r = 0; r < rows; r++) memcpy(out_rgb + (size_t)idx[r] * cam->width * 3, &packed[(size_t)r * cam->width * 3], row_bytes);
    }
    st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->t_begin).count();
    if (stats) *stats = st;
    return MCRT_OK;
}

int mcrt_emit_photons(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, mcrt_photon_emission* out) {
    return mcrt_emit_photons_shard(ctx, emissions, caustic_factor, global_seed, 0, 1, out);
}

}  // extern "C"

namespace {
// The emission pass; the lists stay in ctx->emit_photons / emit_keys. h = the kernel's counters ([1] global, [2] caustic
// photons, [3] paths, [4] rays).
int emitOnDevice(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, uint32_t shard_index, uint32_t shard_count,
                 unsigned long long h[8], float& ms) {
    for (int i = 0; i < 8; i++) h[i] = 0ull;
    ms = 0.f;
    if (shard_count == 0 || shard_index >= shard_count) return fail(ctx, MCRT_ERR_INVALID, "shard_index >= shard_count");
    if (!ctx->has_scene) return fail(ctx, MCRT_ERR_NO_SCENE, "mcrt_emit_photons before mcrt_upload_scene");
    REJECT_IF_PENDING(ctx, "mcrt_emit_photons");
    if (!(emissions >= 0.0) || !(caustic_factor > 0.0)) return fail(ctx, MCRT_ERR_INVALID, "emissions must be >= 0 and caustic_factor > 0");
    const uint32_t nl = ctx->scene.num_lights;
    if (nl == 0 || nl > 0xFFFFu) return nl == 0 ? MCRT_OK : fail(ctx, MCRT_ERR_UNSUPPORTED, "more than 65535 lights");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // work split, photon-mapper.cpp:31-78
    const size_t photon_emissions = (size_t)((double)(size_t)emissions * caustic_factor);
    double total_add_flux = 0.0;
    for (uint32_t i = 0; i < nl; i++) {
        const double* f = &ctx->host_light_flux[(size_t)i * 3];
        total_add_flux += 0.0 + f[0] + f[1] + f[2];  // glm::compAdd
    }
    std::vector<unsigned long long> first(nl + 1, 0ull);
    std::vector<double> pflux((size_t)nl * 3);
    for (uint32_t i = 0; i < nl; i++) {
        const double* f = &ctx->host_light_flux[(size_t)i * 3];
        const double share = (0.0 + f[0] + f[1] + f[2]) / total_add_flux;
        const size_t n = (size_t)((double)photon_emissions * share);
        if (n > 0xFFFFFFFFull) return fail(ctx, MCRT_ERR_UNSUPPORTED, "more than 2^32 emissions from one light");
        first[i + 1] = first[i] + n;
        for (int c = 0; c < 3; c++) pflux[(size_t)i * 3 + c] = f[c] / (double)n;
    }
    const unsigned long long all_paths = first[nl];
    const unsigned long long shard_begin = all_paths * shard_index / shard_count, shard_end = all_paths * (shard_index + 1ull) / shard_count;
    const unsigned long long total = shard_end - shard_begin;  // paths of this shard
    if (int rc = uploadArray(ctx, ctx->emit_first, first.data(), first.size())) return rc;
    if (int rc = uploadArray(ctx, ctx->emit_flux, pflux.data(), pflux.size())) return rc;
    if (!ctx->emit_counters.p) HIP_TRY(ctx, ctx->emit_counters.alloc(8 * sizeof(unsigned long long)));

    ctx->lean_used = false;
    auto kernel = leanOf(ctx, ctx->scene.stage_all ? emitKernel<true> : emitKernel<false>);
    DeviceScene scene = ctx->scene;
    scene.flat = 0;  // the emission kernel walks the BVH
    LaunchGeom g;
    if (int rc = launchGeometry(ctx, kernel, scene, g)) return rc;
    if (int rc = ensureScratch(ctx, g.total_lanes, false)) return rc;

    // List sizes. A path leaves a photon at every diffuse bounce, so the lists can hold more photons than there are paths (C5: 1.13 per
    // path in the caustic list) or far fewer (its global list: 0.07). A launch whose lists are too small still COUNTS exactly, and the pass
    // is repeated at the exact sizes (below) - until round 4 that was the normal case for C5, whose emission so ran twice (0.37 s each).
    // Now a pilot launch over every 64th path of the range (capacity 0: counting only, ~1/64 of the time) sizes the lists first, with 5 %
    // and 64 K photons to spare; lists left by an earlier call are used at their full size.
    unsigned long long cap[2] = {std::max<unsigned long long>(1ull << 16, total), std::max<unsigned long long>(1ull << 16, total)};
    constexpr uint32_t kPilotStride = 64;
    const bool pilot = total >= (4ull << 20);
    for (int attempt = pilot ? -1 : 0; attempt < 3; attempt++) {
        const bool counting = attempt < 0;
        for (int w = 0; w < 2 && !counting; w++) {
            if (ctx->emit_photons[w].bytes < cap[w] * 32) HIP_TRY(ctx, ctx->emit_photons[w].alloc(cap[w] * 32));
            if (ctx->emit_keys[w].bytes < cap[w] * 8) HIP_TRY(ctx, ctx->emit_keys[w].alloc(cap[w] * 8));
            cap[w] = std::min<unsigned long long>(ctx->emit_photons[w].bytes / 32, ctx->emit_keys[w].bytes / 8);
        }
        EmitParams prm;
        memset(&prm, 0, sizeof(prm));
        prm.num_lights = nl;
        prm.light_first = ctx->emit_first.as<unsigned long long>();
        prm.light_photon_flux = ctx->emit_flux.as<double>();
        prm.total_emissions = shard_end;
        prm.first_emission = shard_begin;
        prm.stride = counting ? kPilotStride : 1u;
        prm.global_seed = global_seed;
        prm.non_caustic_reject = 1.0 / caustic_factor;
        for (int w = 0; w < 2; w++) {
            prm.photons[w] = ctx->emit_photons[w].as<float>();
            prm.keys[w] = ctx->emit_keys[w].as<unsigned long long>();
            prm.capacity[w] = counting ? 0ull : cap[w];
        }
        prm.counters = ctx->emit_counters.as<unsigned long long>();
        prm.spill = ctx->spill.as<StackEntry>();
        prm.total_lanes = g.total_lanes;
        const uint32_t grid = (uint32_t)std::min<unsigned long long>(g.grid, (total + kBlock - 1) / kBlock + 1);
        HIP_TRY(ctx, hipMemsetAsync(ctx->emit_counters.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), g.lds_bytes, ctx->stream, scene, prm);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        HIP_TRY(ctx, hipMemcpy(h, ctx->emit_counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        if (h[5]) return fail(ctx, MCRT_ERR_UNSUPPORTED, "traversal stack overflow in the emission pass");
        // RefractionHistory (ray.cpp:74-98) is unbounded in the reference; the emission kernel keeps kMaxIors (8) entries per lane. The eye
        // pass of such a scene retries through the 32-entry pool or fails (mcrt_render_finish); the photon pass must not be the silent one.
        if (h[6]) return fail(ctx, MCRT_ERR_UNSUPPORTED, "a photon path entered more than 8 nested dielectric media (RefractionHistory, ray.cpp:74-98, is kept to 8 entries per lane in the emission pass)");
        if (counting) {
            for (int w = 0; w < 2; w++) cap[w] = (unsigned long long)((double)h[1 + w] * kPilotStride * 1.05) + (1ull << 16);
            continue;
        }
        if (h[1] <= cap[0] && h[2] <= cap[1]) break;
        cap[0] = std::max(cap[0], h[1]);  // a list was too small: size it exactly and emit again
        cap[1] = std::max(cap[1], h[2]);
        if (attempt == 2) return fail(ctx, MCRT_ERR_HIP, "photon lists kept overflowing");
    }
    return MCRT_OK;
}
}  // namespace

extern "C" {

int mcrt_emit_photons_shard(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, uint32_t shard_index,
                            uint32_t shard_count, mcrt_photon_emission* out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!out) return fail(ctx, MCRT_ERR_INVALID, "out is NULL");
    memset(out, 0, sizeof(*out));
    unsigned long long h[8];
    float ms = 0.f;
    if (int rc = emitOnDevice(ctx, emissions, caustic_factor, global_seed, shard_index, shard_count, h, ms)) return rc;
    for (int w = 0; w < 2; w++) {
        const size_t n = (size_t)h[1 + w];
        ctx->host_photons[w].resize(n * 8);
        ctx->host_keys[w].resize(n);
        if (n) {
            HIP_TRY(ctx, hipMemcpy(ctx->host_photons[w].data(), ctx->emit_photons[w].p, n * 32, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(ctx->host_keys[w].data(), ctx->emit_keys[w].p, n * 8, hipMemcpyDeviceToHost));
        }
    }
    out->global_count = h[1];
    out->caustic_count = h[2];
    out->global_photons = ctx->host_photons[0].data();
    out->caustic_photons = ctx->host_photons[1].data();
    out->global_keys = ctx->host_keys[0].data();
    out->caustic_keys = ctx->host_keys[1].data();
    out->emission_paths = h[3];
    out->rays = h[4];
    out->kernel_ms = ms;
    return MCRT_OK;
}

int mcrt_emit_photons_device(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, uint32_t shard_index,
                             uint32_t shard_count, mcrt_photon_emission_device* out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!out) return fail(ctx, MCRT_ERR_INVALID, "out is NULL");
    memset(out, 0, sizeof(*out));
    unsigned long long h[8];
    float ms = 0.f;
    if (int rc = emitOnDevice(ctx, emissions, caustic_factor, global_seed, shard_index, shard_count, h, ms)) return rc;
    out->global_count = h[1];
    out->caustic_count = h[2];
    out->d_global_photons = ctx->emit_photons[0].as<float>();
    out->d_caustic_photons = ctx->emit_photons[1].as<float>();
    out->emission_paths = h[3];
    out->rays = h[4];
    out->kernel_ms = ms;
    return MCRT_OK;
}

// buildMapOnDevice, or - when more than a leaf's worth of photons share one cell of its 21-level codes (2^-21 of the map's cube: the
// focus of a sharp caustic can do that) - the same map from the recursive host builder, which splits as deep as the reference's Octree
// does (octree.cpp:35-80): the list goes to the host once, the finished map comes back (mcrt_upload_photons' path). Slower, and what the
// call used to refuse with MCRT_ERR_UNSUPPORTED until round 4.
int buildMapAnyDepth(mcrt_ctx* ctx, int which, const float* d_photons, uint64_t n, const double bb_min[3], const double bb_max[3],
                     uint32_t max_node_data, double* timing) {
    ctx->dense_cell_refused = false;
    const int rc = buildMapOnDevice(ctx, which, d_photons, n, bb_min, bb_max, max_node_data, timing);
    if (rc != MCRT_ERR_UNSUPPORTED || !ctx->dense_cell_refused) return rc;
    ctx->dense_cell_refused = false;
    std::vector<float> host;
    try {
        host.resize((size_t)n * 8);
    } catch (...) {
        return fail(ctx, MCRT_ERR_HIP, "photon map: out of host memory for the recursive builder's copy of the list");
    }
    HIP_TRY(ctx, hipMemcpy(host.data(), d_photons, (size_t)n * 32, hipMemcpyDeviceToHost));
    mcrt_photon_map* M = nullptr;
    if (int rc2 = mcrt_photon_map_build(host.data(), n, bb_min, bb_max, max_node_data, &M)) return fail(ctx, rc2, "photon map: the recursive host builder failed");
    const int rc3 = uploadMap(ctx, which, mcrt_photon_map_get(M));
    mcrt_photon_map_free(M);
    return rc3;
}

int mcrt_upload_photons_device(mcrt_ctx* ctx, const float* d_global_photons, uint64_t global_count, const float* d_caustic_photons,
                               uint64_t caustic_count, const double bb_min[3], const double bb_max[3], uint32_t max_photons_per_leaf,
                               uint32_t k_nearest_photons, int direct_visualization, mcrt_photon_pass_stats* stats) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (k_nearest_photons == 0 || max_photons_per_leaf == 0 || !bb_min || !bb_max) return fail(ctx, MCRT_ERR_INVALID, "mcrt_upload_photons_device: bad argument");
    if ((global_count && !d_global_photons) || (caustic_count && !d_caustic_photons)) return fail(ctx, MCRT_ERR_INVALID, "mcrt_upload_photons_device: null photon list");
    REJECT_IF_PENDING(ctx, "mcrt_upload_photons_device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    ctx->has_photons = false;
    ctx->k_nearest = k_nearest_photons;  // before the maps: the record lists expand octants with more than k photons
    double timing[3] = {0.0, 0.0, 0.0};
    if (int rc = buildMapAnyDepth(ctx, 0, d_global_photons, global_count, bb_min, bb_max, max_photons_per_leaf, timing)) return rc;
    if (int rc = buildMapAnyDepth(ctx, 1, d_caustic_photons, caustic_count, bb_min, bb_max, max_photons_per_leaf, timing)) return rc;
    ctx->direct_visualization = direct_visualization ? 1 : 0;
    ctx->has_photons = true;
    if (stats) {
        stats->global_count = global_count;
        stats->caustic_count = caustic_count;
        stats->global_octants = ctx->maps[0].num_octants;
        stats->caustic_octants = ctx->maps[1].num_octants;
        stats->sort_ms = timing[0];
        stats->octant_ms = timing[1];
        stats->finish_ms = timing[2];
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return MCRT_OK;
}

int mcrt_photon_pass_device(mcrt_ctx* ctx, double emissions, double caustic_factor, uint32_t global_seed, const double bb_min[3],
                            const double bb_max[3], uint32_t max_photons_per_leaf, uint32_t k_nearest_photons, int direct_visualization,
                            mcrt_photon_pass_stats* stats) {
    if (!ctx) return MCRT_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    mcrt_photon_emission_device em;
    if (int rc = mcrt_emit_photons_device(ctx, emissions, caustic_factor, global_seed, 0, 1, &em)) return rc;
    mcrt_photon_pass_stats st;
    memset(&st, 0, sizeof(st));
    if (int rc = mcrt_upload_photons_device(ctx, em.d_global_photons, em.global_count, em.d_caustic_photons, em.caustic_count, bb_min, bb_max,
                                            max_photons_per_leaf, k_nearest_photons, direct_visualization, &st))
        return rc;
    st.emission_paths = em.emission_paths;
    st.rays = em.rays;
    st.emission_ms = em.kernel_ms;
    st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return MCRT_OK;
}

int mcrt_photon_map_download(mcrt_ctx* ctx, int which, mcrt_photon_map** out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!out || which < 0 || which > 1) return fail(ctx, MCRT_ERR_INVALID, "mcrt_photon_map_download: bad argument");
    if (!ctx->has_photons) return fail(ctx, MCRT_ERR_NO_PHOTONS, "mcrt_photon_map_download before the maps exist");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const PhotonMapView& v = ctx->maps[which];
    mcrt_photon_map* M = nullptr;
    try {  // (host copies of up to GBs: bad_alloc must not cross the C boundary)
    M = new mcrt_photon_map();
    const size_t no = v.num_octants, np = (size_t)v.num_photons;
    if (no) {
        std::vector<uint32_t> start(no), contained(no);
        M->bounds.resize(no * 6);
        M->next.resize(no);
        M->leaf.resize(no);
        M->photons.resize(np * 8);
        hipError_t e = hipMemcpy(M->bounds.data(), v.octant_bounds, no * 48, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(start.data(), v.octant_start, no * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(contained.data(), v.octant_contained, no * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(M->next.data(), v.octant_next, no * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(M->leaf.data(), v.octant_leaf, no, hipMemcpyDeviceToHost);
        if (e == hipSuccess && np) e = hipMemcpy(M->photons.data(), v.photons, np * 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            delete M;
            return fail(ctx, MCRT_ERR_HIP, std::string("mcrt_photon_map_download: ") + hipGetErrorString(e));
        }
        M->start.assign(start.begin(), start.end());
        M->contained.assign(contained.begin(), contained.end());
    }
    finishMapDesc(M);
    } catch (...) {
        delete M;
        return fail(ctx, MCRT_ERR_HIP, "mcrt_photon_map_download: out of host memory");
    }
    *out = M;
    return MCRT_OK;
}

int mcrt_intersect(mcrt_ctx* ctx, uint64_t n, const double* start, const double* direction, double* out_t, uint32_t* out_surface,
                   double* out_uv) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, MCRT_ERR_NO_SCENE, "mcrt_intersect before mcrt_upload_scene");
    REJECT_IF_PENDING(ctx, "mcrt_intersect");
    if (n == 0) return MCRT_OK;
    if (!start || !direction || !out_t || !out_surface) return fail(ctx, MCRT_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->scene.stage_all && ctx->scene.num_nodes > 0 && n <= 0xFFF00000ull) {  // (32-bit queue cursors with room for the waves' overshoot)
        // tree in HBM: the trace kernel of the wavefront pipeline, fed from the arrays
        const int lean = ctxOptL(ctx, "MCRT_WF_LEAN", 1) != 0 ? (ctx->q_single && ctxOptL(ctx, "MCRT_WF_LEAN", 1) != 2 ? 3 : 1) : 0;  // as launchWavefront
        auto trace = lean == 3 ? wfTraceKernel<ArrayRays, false, 3> : lean == 1 ? wfTraceKernel<ArrayRays, false, 1> : wfTraceKernel<ArrayRays, false>;
        TracePlan tp;
        if (int rc = planTrace(ctx, trace, n, tp)) return rc;
        DevBuf &ds = ctx->op_buf[0], &dd = ctx->op_buf[1], &dt = ctx->op_buf[2], &dsf = ctx->op_buf[3], &duv = ctx->op_buf[4];
        if (int rc = uploadInto(ctx, ds, start, n * 3)) return rc;
        if (int rc = uploadInto(ctx, dd, direction, n * 3)) return rc;
        HIP_TRY(ctx, dt.reserve(n * 8));
        HIP_TRY(ctx, dsf.reserve(n * 4));
        HIP_TRY(ctx, duv.reserve(n * 16));
        if (!ctx->wf_ctrl.p) HIP_TRY(ctx, ctx->wf_ctrl.alloc(kWfCtrlWords * sizeof(unsigned long long)));
        const unsigned long long ctrl_init[4] = {n, 0ull, 0ull, 0ull};
        HIP_TRY(ctx, hipMemcpyAsync(ctx->wf_ctrl.p, ctrl_init, sizeof(ctrl_init), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->stats.p, 0, kStatsWords * sizeof(unsigned long long), ctx->stream));
        WfTraceArgs ta = tp.args;
        ta.count = ctx->wf_ctrl.as<unsigned long long>();
        ta.pop = ctx->wf_ctrl.as<unsigned long long>() + 2;
        ArrayRays ar{ds.as<double>(), dd.as<double>(), dt.as<double>(), dsf.as<uint32_t>(), duv.as<double>()};
        const bool op_time = ctxOptOn(ctx, "MCRT_OP_TIME");  // kernel time of the operator to stderr
        if (op_time) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        hipLaunchKernelGGL(trace, dim3(tp.grid), dim3(tp.block), tp.lds_bytes, ctx->stream, ta, ar);
        HIP_TRY(ctx, hipGetLastError());
        if (op_time) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (op_time) {
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            fprintf(stderr, "[mcrt op] intersect (trace kernel): %llu rays in %.3f ms = %.1f Mray/s\n", (unsigned long long)n, ms, n / (ms * 1e3));
        }
        unsigned long long h[kStatsWords];
        HIP_TRY(ctx, hipMemcpy(h, ctx->stats.p, sizeof(h), hipMemcpyDeviceToHost));
        if (h[5]) return fail(ctx, MCRT_ERR_UNSUPPORTED, "traversal stack overflow (internal error: the stacks are sized to the tree's own bound, HostLayout::stack_bound)");
        HIP_TRY(ctx, hipMemcpy(out_t, dt.p, n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(out_surface, dsf.p, n * 4, hipMemcpyDeviceToHost));
        if (out_uv) HIP_TRY(ctx, hipMemcpy(out_uv, duv.p, n * 16, hipMemcpyDeviceToHost));
        return MCRT_OK;
    }
    LaunchGeom g;
    auto ikernel = ctx->scene.stage_all ? intersectKernel<true> : intersectKernel<false>;
    if (int rc = launchGeometry(ctx, ikernel, ctx->scene, g)) return rc;
    if (int rc = ensureScratch(ctx, g.total_lanes, false)) return rc;
    DevBuf &ds = ctx->op_buf[0], &dd = ctx->op_buf[1], &dt = ctx->op_buf[2], &dsf = ctx->op_buf[3], &duv = ctx->op_buf[4];
    if (int rc = uploadInto(ctx, ds, start, n * 3)) return rc;
    if (int rc = uploadInto(ctx, dd, direction, n * 3)) return rc;
    HIP_TRY(ctx, dt.reserve(n * 8));
    HIP_TRY(ctx, dsf.reserve(n * 4));
    HIP_TRY(ctx, duv.reserve(n * 16));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(g.grid, (n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ikernel, dim3(grid), dim3(kBlock), g.lds_bytes, ctx->stream, ctx->scene, n, ds.as<double>(),
                       dd.as<double>(), dt.as<double>(), dsf.as<uint32_t>(), duv.as<double>(), ctx->spill.as<StackEntry>(),
                       g.total_lanes);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out_t, dt.p, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out_surface, dsf.p, n * 4, hipMemcpyDeviceToHost));
    if (out_uv) HIP_TRY(ctx, hipMemcpy(out_uv, duv.p, n * 16, hipMemcpyDeviceToHost));
    return MCRT_OK;
}

int mcrt_sampler(mcrt_ctx* ctx, uint64_t n, const uint32_t* pixel, const uint32_t* index, uint32_t shuffles, uint32_t global_seed,
                 double* out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (n == 0) return MCRT_OK;
    if (!pixel || !index || !out) return fail(ctx, MCRT_ERR_INVALID, "null argument");
    REJECT_IF_PENDING(ctx, "mcrt_sampler");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf &dp = ctx->op_buf[0], &di = ctx->op_buf[1], &dout = ctx->op_buf[2];
    if (int rc = uploadInto(ctx, dp, pixel, n)) return rc;
    if (int rc = uploadInto(ctx, di, index, n)) return rc;
    HIP_TRY(ctx, dout.reserve(n * 7 * 8));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(samplerKernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->sobol_tab.as<uint32_t>(), n, dp.as<uint32_t>(),
                       di.as<uint32_t>(), shuffles, global_seed, dout.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 7 * 8, hipMemcpyDeviceToHost));
    return MCRT_OK;
}

int mcrt_bsdf(mcrt_ctx* ctx, uint64_t n, const double* in, const double* consts, double* out) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (n == 0) return MCRT_OK;
    if (!in || !consts || !out) return fail(ctx, MCRT_ERR_INVALID, "null argument");
    REJECT_IF_PENDING(ctx, "mcrt_bsdf");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BsdfKatConsts c;
    memset(&c, 0, sizeof(c));
    c.rough.roughness = consts[0];
    for (int k = 0; k < 3; k++) {
        c.rough.reflectance[k] = consts[1 + k];
        c.real[k] = consts[4 + k];
        c.imag[k] = consts[7 + k];
    }
    const double variance = c.rough.roughness * c.rough.roughness;  // Material::computeProperties, material/material.cpp:106-108
    c.rough.A = 1.0 - 0.5 * (variance / (variance + 0.33));
    c.rough.B = 0.45 * (variance / (variance + 0.09));
    c.rough.flags = MCRT_MAT_ROUGH;
    DevBuf &din = ctx->op_buf[0], &dout = ctx->op_buf[1];
    if (int rc = uploadInto(ctx, din, in, n * 11)) return rc;
    HIP_TRY(ctx, dout.reserve(n * 18 * 8));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(bsdfKernel, dim3(grid), dim3(256), 0, ctx->stream, n, din.as<double>(), c, dout.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 18 * 8, hipMemcpyDeviceToHost));
    return MCRT_OK;
}

int mcrt_libm(mcrt_ctx* ctx, int fn, uint64_t n, const double* a, const double* b, double* out0, double* out1) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (fn < MCRT_LIBM_SINCOS || fn > MCRT_LIBM_POW) return fail(ctx, MCRT_ERR_INVALID, "mcrt_libm: unknown function selector");
    if (n == 0) return MCRT_OK;
    const bool two = fn == MCRT_LIBM_SINCOS || fn == MCRT_LIBM_SINCOSF, pair = fn == MCRT_LIBM_ATAN2 || fn == MCRT_LIBM_POW;
    if (!a || !out0 || (pair && !b) || (two && !out1)) return fail(ctx, MCRT_ERR_INVALID, "null argument");
    REJECT_IF_PENDING(ctx, "mcrt_libm");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf &da = ctx->op_buf[0], &db = ctx->op_buf[1], &d0 = ctx->op_buf[2], &d1 = ctx->op_buf[3];
    if (int rc = uploadInto(ctx, da, a, n)) return rc;
    if (pair)
        if (int rc = uploadInto(ctx, db, b, n)) return rc;
    HIP_TRY(ctx, d0.reserve(n * 8));
    HIP_TRY(ctx, d1.reserve(n * 8));
    if (fn == MCRT_LIBM_POW) {  // (the output stage's function: its kernel lives with that stage, mcrt_output.hip)
        HIP_TRY(ctx, (hipError_t)launchPowKat(ctx->stream, n, da.as<double>(), db.as<double>(), d0.as<double>()));
    } else {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(4096, (n + 255) / 256);
        hipLaunchKernelGGL(libmKernel, dim3(grid), dim3(256), 0, ctx->stream, fn, n, da.as<double>(), db.as<double>(), d0.as<double>(), d1.as<double>());
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out0, d0.p, n * 8, hipMemcpyDeviceToHost));
    if (two) HIP_TRY(ctx, hipMemcpy(out1, d1.p, n * 8, hipMemcpyDeviceToHost));
    return MCRT_OK;
}

int mcrt_knn(mcrt_ctx* ctx, int which, uint64_t n, const double* p, uint32_t k, uint32_t* out_count, uint32_t* out_index,
             double* out_distance2) {
    if (!ctx) return MCRT_ERR_INVALID;
    if (!ctx->has_photons) return fail(ctx, MCRT_ERR_NO_PHOTONS, "mcrt_knn before mcrt_upload_photons");
    REJECT_IF_PENDING(ctx, "mcrt_knn");
    if (which < 0 || which > 1 || k == 0) return fail(ctx, MCRT_ERR_INVALID, "bad map selector or k");
    if (n == 0) return MCRT_OK;
    if (!p || !out_count || !out_index || !out_distance2) return fail(ctx, MCRT_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const char* kenv = ctxOpt(ctx, "MCRT_KERNEL");
    if (k <= waveMaxK(kWaveRowsLarge) && !(kenv && strcmp(kenv, "legacy") == 0)) {  // wave-cooperative search (mcrt_waveknn.hpp)
        DevBuf &dp = ctx->op_buf[0], &dc = ctx->op_buf[1], &di = ctx->op_buf[2], &dd = ctx->op_buf[3], &flags = ctx->op_buf[4];
        if (int rc = uploadInto(ctx, dp, p, n * 3)) return rc;
        HIP_TRY(ctx, dc.reserve(n * 4));
        HIP_TRY(ctx, di.reserve(n * k * 4));
        HIP_TRY(ctx, dd.reserve(n * k * 8));
        HIP_TRY(ctx, flags.reserve(8));
        HIP_TRY(ctx, hipMemsetAsync(flags.p, 0, 8, ctx->stream));
        const PhotonMapViewW mv = waveMapView(ctx, which);
        HIP_TRY(ctx, ctx->knn_spill.reserve((size_t)ctx->num_cus * std::max(1, (int)ctxOptL(ctx, "MCRT_KNN_BLOCKS", 8)) * 4 * kWaveSpill * 12));
        // MCRT_KNN_BLOCKS: 256-lane workgroups per CU (occupancy experiments); MCRT_KNN_TIME=1: kernel time on stderr
        const int per_cu = std::max(1, (int)ctxOptL(ctx, "MCRT_KNN_BLOCKS", 8));
        const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cus * per_cu, (n + 3) / 4);
        // MCRT_KNN_GROUPS=1: four queries per wave, one per row of 16 lanes (mcrt_groupknn.hpp)
        const bool groups = k <= kGrpMaxK && ctxOptOn(ctx, "MCRT_KNN_GROUPS");
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        if (groups)
            hipLaunchKernelGGL(knnGroupKernel, dim3(std::min<uint32_t>(grid, (uint32_t)((n + 15) / 16))), dim3(256), 0, ctx->stream, mv, n, dp.as<double>(), k,
                               dc.as<uint32_t>(), di.as<uint32_t>(), dd.as<double>(), flags.as<unsigned long long>());
        else
            hipLaunchKernelGGL((k > waveMaxK(kWaveRows) ? knnWaveKernel<kWaveRowsLarge> : knnWaveKernel<kWaveRows>), dim3(grid), dim3(256), 0, ctx->stream, mv, n,
                               dp.as<double>(), k, dc.as<uint32_t>(), di.as<uint32_t>(), dd.as<double>(), flags.as<unsigned long long>(),
                               ctx->knn_spill.as<uint32_t>());
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctxOptOn(ctx, "MCRT_KNN_TIME")) {
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            fprintf(stderr, "[mcrt knn] map %d: %llu searches, k = %u, %d workgroups per CU: %.3f ms = %.1f M searches/s\n", which,
                    (unsigned long long)n, k, per_cu, ms, (double)n / ms / 1e3);
        }
        unsigned long long f = 0;
        HIP_TRY(ctx, hipMemcpy(&f, flags.p, 8, hipMemcpyDeviceToHost));
        if (ctxOptOn(ctx, "MCRT_TEST_KNN_OVERFLOW")) f = 1;  // (test hook: take the branch below whatever the searches did)
        if (!f) {
            HIP_TRY(ctx, hipMemcpy(out_count, dc.p, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(out_index, di.p, n * k * 4, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(out_distance2, dd.p, n * k * 8, hipMemcpyDeviceToHost));
            return MCRT_OK;
        }
        // a search had more octants pending at once than the wave's frontier holds (128 in registers + a 1 024-entry list in memory for the
        // wave-per-query kernel; 128 for the four-queries-per-wave kernel of small k): the call is served by the per-lane kernel below,
        // whose frontier grows (the reference's queue is unbounded, linear-octree.cpp:33; until round 6: MCRT_ERR_UNSUPPORTED)
        ctx->knn_visit_cap = std::max<uint32_t>(ctx->knn_visit_cap, 2048u);
    }
    const uint32_t block = 64;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)ctx->num_cus * 8, (n + block - 1) / block);
    const uint32_t lanes = grid * block;
    DevBuf dp, dc, di, dd, r_d2, r_idx, v_d2, v_oct, lane_flag;
    if (int rc = uploadArray(ctx, dp, p, n * 3)) return rc;
    HIP_TRY(ctx, dc.alloc(n * 4));
    HIP_TRY(ctx, di.alloc(n * k * 4));
    HIP_TRY(ctx, dd.alloc(n * k * 8));
    HIP_TRY(ctx, r_d2.alloc((size_t)lanes * k * 8));
    HIP_TRY(ctx, r_idx.alloc((size_t)lanes * k * 4));
    HIP_TRY(ctx, lane_flag.alloc(8));
    for (;;) {  // (the per-lane frontier: 160 entries per lane to begin with, eight times as many whenever a search ran out)
        const uint32_t cap = ctx->knn_visit_cap;
        HIP_TRY(ctx, v_d2.alloc((size_t)lanes * cap * 8));
        HIP_TRY(ctx, v_oct.alloc((size_t)lanes * cap * 4));
        HIP_TRY(ctx, hipMemsetAsync(lane_flag.p, 0, 8, ctx->stream));
        hipLaunchKernelGGL(knnKernel, dim3(grid), dim3(block), 0, ctx->stream, ctx->maps[which], n, dp.as<double>(), k, dc.as<uint32_t>(),
                           di.as<uint32_t>(), dd.as<double>(), r_d2.as<double>(), r_idx.as<uint32_t>(), v_d2.as<double>(),
                           v_oct.as<uint32_t>(), lanes, cap, lane_flag.as<unsigned long long>());
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long f = 0;
        HIP_TRY(ctx, hipMemcpy(&f, lane_flag.p, 8, hipMemcpyDeviceToHost));
        if (!f) break;
        if (cap >= kMaxVisitLimit)
            return fail(ctx, MCRT_ERR_UNSUPPORTED, "kNN frontier overflow: a search had more than " + std::to_string(kMaxVisitLimit) + " octants pending at once (per-lane kernel)");
        ctx->knn_visit_cap = std::min<uint32_t>(cap * 8u, kMaxVisitLimit);
    }
    HIP_TRY(ctx, hipMemcpy(out_count, dc.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out_index, di.p, n * k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out_distance2, dd.p, n * k * 8, hipMemcpyDeviceToHost));
    return MCRT_OK;
}

}  // extern "C"

namespace mcrt {
int ctxDevice(const mcrt_ctx* ctx) { return ctx->device; }
void* ctxStream(const mcrt_ctx* ctx) { return (void*)ctx->stream; }
int ctxFail(mcrt_ctx* ctx, int code, const std::string& msg) { return fail(ctx, code, msg); }
}  // namespace mcrt
