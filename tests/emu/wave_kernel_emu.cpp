// wfTraceKernel itself (csrc/mcrt_kernels.hpp: the persistent-wave trace kernel of the wavefront pipeline and of mcrt_intersect -
// queue dealt in blocks to the workgroups, LDS cursor, batched refills and hit stores, the gates, the top of the tree and the root
// staged in LDS, per-lane stacks in LDS + spill) run on the HOST: workgroups of several emulated wavefronts (wave_emu.hpp:
// cross-lane operations per wave, __syncthreads per workgroup), one workgroup after the other, the device source unchanged.
// Arguments are filled the way planTrace (mcrt_hip.hip) fills them. Test harness only.
#define MCRT_WAVE_EMU 1
#include "wave_emu.hpp"

#include "mcrt_emu.cpp"

#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_groupknn.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_widerec.hpp"

namespace {
alignas(64) unsigned char lds[160 * 1024];  // what `extern __shared__ unsigned char lds[]` of the kernels refers to here
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_kernels.hpp"

// a map of the host desc as the wave search wants it (record lists: buildWideRecords, positions by themselves)
struct WaveMap {
    std::vector<uint32_t> start, contained;
    std::vector<WideRec> wide;
    std::vector<PhotonPos> pos;
    PhotonMapViewW view;
    int init(const mcrt_photon_map_desc* m, uint32_t k) {
        memset(&view, 0, sizeof(view));
        if (!m || m->num_octants == 0) return 0;
        const size_t no = m->num_octants;
        start.resize(no);
        contained.resize(no);
        for (size_t i = 0; i < no; i++) {
            start[i] = (uint32_t)m->octant_start_data[i];
            contained[i] = (uint32_t)m->octant_contained_data[i];
        }
        uint32_t ra = 0, rm = 0;
        if (const int rc = buildWideRecords(m, contained.data(), k ? k : 1u, wide, ra, rm)) return rc;
        pos.resize((size_t)m->num_photons);
        for (size_t i = 0; i < pos.size(); i++) pos[i] = PhotonPos{m->photons[8 * i + 3], m->photons[8 * i + 4], m->photons[8 * i + 5]};
        view.base.num_octants = m->num_octants;
        view.base.num_photons = m->num_photons;
        view.base.octant_bounds = m->octant_bounds;
        view.base.octant_start = start.data();
        view.base.octant_contained = contained.data();
        view.base.octant_next = m->octant_next_sibling;
        view.base.octant_leaf = m->octant_leaf;
        view.base.photons = m->photons;
        view.wide = wide.data();
        view.root_a = ra;
        view.root_m = rm;
        view.pos = pos.data();
        return 0;
    }
};

template <int kLean>
void launchTrace(const WfTraceArgs& a, const ArrayRays& rays, uint32_t grid, uint32_t waves) {
    for (uint32_t g = 0; g < grid; g++) {
        wemu::launch().block_idx = g;
        wemu::launch().block_dim = waves * 64u;
        wemu::launch().grid_dim = grid;
        wemu::runGroup((int)waves, [&](int) { wfTraceKernel<ArrayRays, true, kLean>(a, rays); });
    }
}
}  // namespace

extern "C" {

// Closest hits of n rays through wfTraceKernel<ArrayRays, count, lean> (form: 3 = round 4's visit, 11 = the lean visit, 27 = the lean visit
// with one block per visit; the forms 0 / 1 / 2 of rounds 2-3 were removed in round 6: -201). grid workgroups of `waves` wavefronts; lds_blocks / lds_stack / refill /
// leaf_lanes / deal_shift as planTrace's options (lds_blocks 0xFFFFFFFF = as many as the tree has, up to 512). stats: the kernel's
// counters [kStatsWords] (rays at [1], node / primitive tests at [2] / [3], overflow at [5]). Returns 0, -100 on stack overflow.
int wemu_trace_kernel(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int form, uint32_t grid, uint32_t waves,
                      uint32_t lds_blocks, int lds_stack, int refill_lanes, int leaf_lanes, uint32_t deal_shift, double* out_t, uint32_t* out_surface,
                      double* out_uv, unsigned long long* stats_out) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    if (scene->num_nodes == 0 || grid == 0 || waves == 0 || waves > 16) return -200;
    if (form == 0 || form == 1 || form == 2) return -201;
    const uint32_t block = waves * 64u;
    const uint32_t depth = (uint32_t)std::max<int>(kMaxStackDepth, (int)E.L.stack_bound + 1);
    std::vector<SmStackEntry> spill((size_t)grid * block * depth);
    std::vector<unsigned long long> stats(kStatsWords + 32, 0ull);
    unsigned long long count = n, pop = 0;
    WfTraceArgs a;
    memset(&a, 0, sizeof(a));
    a.count = &count;
    a.pop = &pop;
    a.stats = stats.data();
    a.nodes = E.L.nodes64.data();
    a.qblocks = E.L.qblocks.data();
    a.num_nodes = (uint32_t)E.L.nodes64.size();
    a.lds_blocks = std::min<uint32_t>(lds_blocks, std::min<uint32_t>((uint32_t)E.L.qblocks.size(), 512u));
    a.q_root_a = E.L.q_root_a;
    a.q_root_m = E.L.q_root_m;
    a.prim = E.L.prim.data();
    a.spill = spill.data();
    a.total_lanes = grid * block;
    a.refill_lanes = refill_lanes;
    a.leaf_lanes = leaf_lanes;
    a.leaf_items = 1 << 20;
    a.min_inner = 8;
    a.lds_stack = lds_stack;
    a.max_stack = depth;
    a.deal_shift = deal_shift;
    const size_t lds_need = (size_t)a.lds_blocks * 64u + (size_t)lds_stack * block * sizeof(SmStackEntry) + 64u + waves * kShareMapBytes + 64u;
    if (lds_need > sizeof(lds)) return -202;
    ArrayRays rays;
    rays.start = start;
    rays.direction = direction;
    rays.out_t = out_t;
    rays.out_surface = out_surface;
    rays.out_uv = out_uv;
    if (form == 11) launchTrace<1>(a, rays, grid, waves);  // the lean visit (MCRT_WF_LEAN), any tree
    else if (form == 27) {                                          // ... one block per visit: trees without a node of more than four children
        if (!E.L.q_single) return -203;
        launchTrace<3>(a, rays, grid, waves);
    } else launchTrace<0>(a, rays, grid, waves);
    if (stats_out) memcpy(stats_out, stats.data(), kStatsWords * sizeof(unsigned long long));
    return stats[5] ? -100 : 0;
}

// ---- whole frames ------------------------------------------------------------------------------------------------------------------
// The frame kernels of launchRender (mcrt_hip.hip) - renderKernel<path tracer, flat>, renderKernelSM, renderKernelPM - and
// sampleResolveKernel on emulated workgroups: DeviceScene filled from the host layout the way mcrt_upload_scene fills it (same staging
// rule), RenderParams / PmExtra the way launchRender fills them (same LDS plans, same work units), one pass over the whole frame.
}  // extern "C"

namespace {

constexpr uint32_t kEmuMaxLds = 160u * 1024u - 3520u;  // the device's LDS less the shading kernels' static table (mcrt_create)

void fillDeviceScene(const mcrt_scene_desc* s, Emu& E, DeviceScene& d, uint32_t flat_max) {
    HostLayout& L = E.L;
    memset(&d, 0, sizeof(d));
    d.num_nodes = s->num_nodes;
    d.num_surfaces = s->num_surfaces;
    d.num_materials = s->num_materials;
    d.num_lights = s->num_lights;
    d.node_bounds = L.node_bounds.data();
    d.node_meta = L.node_meta.data();
    d.nodes64 = L.nodes64.data();
    d.qblocks = L.qblocks.data();
    d.num_qblocks = (uint32_t)L.qblocks.size();
    d.q_nodes = (uint32_t)L.nodes64.size();
    d.stack_depth = std::max<uint32_t>((uint32_t)kMaxStackDepth, L.stack_bound + 1u);
    d.q_root_a = L.q_root_a;
    d.q_root_m = L.q_root_m;
    d.prim = L.prim.data();
    d.flat_prim = L.flat_prim.data();
    d.flat_index = L.flat_index.data();
    d.flat_tris = L.flat_tris;
    d.flat_pre = L.flat_pre.empty() ? nullptr : L.flat_pre.data();
    d.pre_tri_pairs = L.pre_tri_pairs;
    d.pre_sph_pairs = L.pre_sph_pairs;
    for (int c = 0; c < 3; c++) d.pre_centre[c] = L.pre_centre[c];
    d.pre_bound = L.pre_bound;
    d.surf_v = L.num_quadric_surfaces ? L.surf_v_patched.data() : s->surf_v;
    d.surf_normal = L.normal.data();
    d.surf_rec = L.shade_rec.data();
    d.surf_vn = s->surf_vn;
    d.surf_area = s->surf_area;
    d.surf_material = s->surf_material;
    d.surf_kind = s->surf_kind;
    d.materials = s->materials;
    d.light_surface = s->light_surface;
    d.light_cdf = s->light_cdf;
    d.sobol_tab = E.tab.data();
    d.scene_ior = s->scene_ior;
    // staging plan, as mcrt_upload_scene
    d.stage_all = 1;
    d.stage_nodes = 0;
    const uint32_t fixed = planLds(DeviceScene{}, kBlock).total;
    if (planLds(d, kBlock).total - fixed > 48u * 1024u || planLds(d, kBlock).total > kEmuMaxLds || L.num_quadric_surfaces) {
        d.stage_all = 0;
        d.stage_nodes = std::min<uint32_t>(d.num_nodes, 512u);
    }
    d.flat = (d.stage_all && d.num_surfaces <= flat_max && !L.flat_prim.empty() && L.num_quadric_surfaces == 0) ? 1u : 0u;
}

template <class F>
void launchGrid(uint32_t grid, uint32_t block, F&& kernel_call) {
    for (uint32_t g = 0; g < grid; g++) {
        wemu::launch().block_idx = g;
        wemu::launch().block_dim = block;
        wemu::launch().grid_dim = grid;
        wemu::runGroup((int)(block / 64u), [&](int) { kernel_call(); }, 1u << 20);
    }
}

}  // namespace

extern "C" {

// A frame by the kernel launchRender picks for it: integrator 0 path tracer / 1 photon mapper; kernel_out: 1 flat, 3 lane state machine,
// 5 photon-mapping wave kernel, 2 wave-synchronous (MCRT_KERNEL_* of include/mcrt.h). force: 0 = launchRender's choice, 2 = the
// wave-synchronous kernel for path-traced frames (MCRT_KERNEL=legacy). grid: workgroups launched (the first takes what work it can).
// out_rgb [h][w][3]; stats_out [kStatsWords]. Returns 0, or a negative code (-100 stack overflow, -202 LDS plan too large, ...).
int wemu_render(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, int force, uint32_t grid,
                double* out_rgb, unsigned long long* stats_out, int* kernel_out) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    DeviceScene d;
    fillDeviceScene(scene, E, d, 64u);
    const bool photon = integrator == MCRT_INTEGRATOR_PHOTON_MAPPER, all = d.stage_all != 0;
    const bool flat_only = !photon && d.flat && force != 2;
    const bool use_sm = !photon && !d.flat && force != 2;
    const bool use_pm_wave = photon && k_nearest <= waveMaxK(kWaveRows);
    if (photon && !use_pm_wave) return -203;
    if (grid == 0) grid = 1;
    RenderParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.cam = *cam;
    prm.global_seed = global_seed;
    prm.spp = cam->sqrtspp * cam->sqrtspp;
    prm.owned_rows = cam->height;
    prm.tiles_x = (cam->width + 7) / 8;
    prm.tiles_y = (prm.owned_rows + 7) / 8;
    unsigned long long work_counter = 0;
    std::vector<unsigned long long> stats(kStatsWords + 32, 0ull);
    prm.work_counter = &work_counter;
    prm.stats = stats.data();
    prm.sm_shade_lanes = 40;
    prm.sm_regen_lanes = 16;
    prm.sm_min_trav = 20;
    prm.sm_leaf_lanes = 32;
    prm.sm_min_inner = 8;
    prm.sm_lds_depth = kLdsStackDepth;
    DeviceScene launch_scene = d;
    uint32_t block = kBlock, lds_bytes = 0, pm_stack_depth = kLdsStackDepth;
    WaveMap wg, wc;
    PmExtra pmx;
    memset(&pmx, 0, sizeof(pmx));
    if (use_pm_wave) {
        if (wg.init(gmap, k_nearest) || wc.init(cmap, k_nearest)) return -301;
        if (!launch_scene.stage_all) launch_scene.stage_nodes = std::min<uint32_t>(launch_scene.stage_nodes, 128u);
        auto ldsBytes = [&](uint32_t b, uint32_t depth) {
            return alignUp(planLds(launch_scene, b, true, depth, b != 1024u ? (uint32_t)kMaxIors : kPmLdsIors).total, 16) +
                   (b / 64) * (waveKnnBytes(kWaveRows) + (all ? 0u : kWaveStateBytes));
        };
        if (launch_scene.flat && ldsBytes(1024, kLdsStackDepth) <= kEmuMaxLds) {
            block = 1024;
        } else if (!launch_scene.stage_all) {
            for (uint32_t depth = 16u; depth >= 2 && block == kBlock; depth -= 2)
                if (ldsBytes(1024, depth) <= kEmuMaxLds) {
                    block = 1024;
                    pm_stack_depth = depth;
                }
        }
        lds_bytes = ldsBytes(block, pm_stack_depth);
    } else if (use_sm) {
        const uint32_t fixed = planSmLds(DeviceScene{}, block, (uint32_t)kLdsStackDepth).total;
        if (!launch_scene.stage_all && fixed < kEmuMaxLds) launch_scene.stage_nodes = std::min<uint32_t>(launch_scene.stage_nodes, (kEmuMaxLds - fixed) / 64u);
        lds_bytes = planSmLds(launch_scene, block, (uint32_t)kLdsStackDepth).total;
    } else {
        lds_bytes = planLds(launch_scene, block, !flat_only).total;
    }
    if (lds_bytes > kEmuMaxLds || lds_bytes > sizeof(lds)) return -202;
    const uint32_t total_lanes = grid * block;
    prm.total_lanes = total_lanes;
    std::vector<StackEntry> spill((size_t)total_lanes * (d.stack_depth - kLdsStackDepth) + 16);
    prm.spill = spill.data();
    std::vector<double> stage, pm_iors;
    std::vector<uint32_t> knn_spill;
    if (use_pm_wave) {
        prm.global_map = wg.view.base;
        prm.caustic_map = wc.view.base;
        prm.k_nearest = k_nearest;
        prm.direct_visualization = direct_visualization ? 1u : 0u;
        pmx.global_map = wg.view;
        pmx.caustic_map = wc.view;
        pmx.stack_depth = pm_stack_depth;
        stage.resize((size_t)kStageDoubles * total_lanes);
        pmx.stage = stage.data();
        knn_spill.resize((size_t)(total_lanes / 64) * kWaveSpill * 3);
        pmx.knn_spill = knn_spill.data();
        if (block == 1024u) {
            pm_iors.resize((size_t)kMaxIors * total_lanes);
            pmx.iors_global = pm_iors.data();
        }
    }
    prm.row_base = 0;
    prm.row_end = prm.owned_rows;
    prm.pass_pixels = (uint64_t)prm.owned_rows * cam->width;
    const ChunkPlan cp = photon ? planChunks(prm.spp, unitsWanted(total_lanes, 128, prm.pass_pixels)) : planChunksMega(prm.spp, total_lanes, prm.pass_pixels);
    prm.chunk_shift = cp.shift;
    prm.chunk = cp.chunk;
    const uint64_t tiles = (uint64_t)prm.tiles_x * ((prm.row_end - prm.row_base + 7) / 8);
    prm.work_items = (tiles * 64ull) << cp.shift;
    std::vector<double> samples((size_t)prm.spp * prm.pass_pixels * 3, 0.0);
    prm.samples = samples.data();
    int kernel_id = 0;
    if (use_pm_wave) {
        kernel_id = 5;
        if (block == 1024u) {
            if (all) launchGrid(grid, block, [&] { renderKernelPM<false, true, 1024>(launch_scene, prm, pmx); });
            else launchGrid(grid, block, [&] { renderKernelPM<false, false, 1024>(launch_scene, prm, pmx); });
        } else {
            if (all) launchGrid(grid, block, [&] { renderKernelPM<false, true>(launch_scene, prm, pmx); });
            else launchGrid(grid, block, [&] { renderKernelPM<false, false>(launch_scene, prm, pmx); });
        }
    } else if (use_sm) {
        kernel_id = 3;
        if (all) launchGrid(grid, block, [&] { renderKernelSM<false, true>(launch_scene, prm); });
        else launchGrid(grid, block, [&] { renderKernelSM<false, false>(launch_scene, prm); });
    } else if (flat_only) {
        kernel_id = 1;
        if (force == 6) {  // the cull records as a kernel argument (renderKernelFlatK, MCRT_FLAT_KARG)
            const size_t floats = (size_t)launch_scene.pre_tri_pairs * kTriPairFloats + (size_t)launch_scene.pre_sph_pairs * kSphPairFloats;
            if (!launch_scene.flat_pre || floats > kFlatPreArgFloats || floats != E.L.flat_pre.size()) return -206;
            FlatPreArg pre;
            memset(&pre, 0, sizeof(pre));
            memcpy(pre.v, E.L.flat_pre.data(), floats * sizeof(float));
            kernel_id = 16;
            launchGrid(grid, block, [&] { renderKernelFlatK<>(launch_scene, prm, pre); });
        } else {
            launchGrid(grid, block, [&] { renderKernel<MCRT_INTEGRATOR_PATH_TRACER, false, true, false, 1>(launch_scene, prm); });
        }
    } else {
        kernel_id = 2;
        if (all) launchGrid(grid, block, [&] { renderKernel<MCRT_INTEGRATOR_PATH_TRACER, false, true>(launch_scene, prm); });
        else launchGrid(grid, block, [&] { renderKernel<MCRT_INTEGRATOR_PATH_TRACER, false, false>(launch_scene, prm); });
    }
    launchGrid((uint32_t)((prm.pass_pixels + 255) / 256), 256, [&] { sampleResolveKernel(prm.samples, prm.pass_pixels, prm.spp, out_rgb); });
    if (stats_out) memcpy(stats_out, stats.data(), kStatsWords * sizeof(unsigned long long));
    if (kernel_out) *kernel_out = kernel_id;
    return stats[5] ? -100 : stats[7] ? -101 : 0;
}

// The wavefront pipeline - wfShadeKernel, wfTraceKernel<PoolRays>, for photon-mapped frames wfKnnKernel<eval>, then sampleResolveKernel -
// launched the way launchWavefront (mcrt_hip.hip) launches them: slot pool and ray queue in (host) memory, control words, a shade
// launch and a trace launch per iteration until a shade launch queues nothing. One pass, box filter, one stream. `slots`: pool slots
// (a multiple of 256 is made of it); trace_grid x trace_waves: the trace launches' shape; trace_form as wemu_trace_kernel.
// launches_out: kernel launches of the frame. The pool starts as garbage except for the planes the device clears too.
int wemu_render_pipeline(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                         int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, uint32_t slots_wanted,
                         uint32_t trace_grid, uint32_t trace_waves, int trace_form, double* out_rgb, unsigned long long* stats_out,
                         uint32_t* launches_out) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    DeviceScene d;
    fillDeviceScene(scene, E, d, 64u);
    if (d.q_nodes == 0 || trace_grid == 0 || trace_waves == 0 || trace_waves > 16) return -200;
    const bool photon = integrator == MCRT_INTEGRATOR_PHOTON_MAPPER;
    if (photon && k_nearest > waveMaxK(kWaveRows)) return -203;
    WfFrame fr;
    memset(&fr, 0, sizeof(fr));
    fr.cam = *cam;
    fr.global_seed = global_seed;
    fr.spp = cam->sqrtspp * cam->sqrtspp;
    fr.tiles_x = (cam->width + 7) / 8;
    fr.film.type = MCRT_FILM_BOX;
    const uint32_t owned_rows = cam->height;
    const uint64_t pixels = (uint64_t)cam->width * owned_rows;
    const uint64_t slots = std::max<uint64_t>((slots_wanted + kWfBlock - 1) / kWfBlock * kWfBlock, kWfBlock);
    {
        const ChunkPlan cp = planChunks(fr.spp, unitsWanted(slots, 16, pixels));
        fr.chunk_shift = cp.shift;
        fr.chunk = cp.chunk;
    }
    std::vector<double> samples((size_t)fr.spp * pixels * 3, 0.0);
    fr.samples = samples.data();
    fr.row_base = 0;
    fr.row_end = owned_rows;
    fr.pass_pixels = pixels;
    fr.work_items = ((unsigned long long)fr.tiles_x * ((owned_rows + 7) / 8) * 64ull) << fr.chunk_shift;
    std::vector<unsigned long long> pool((size_t)slots * kWfWords, 0xDEADBEEFCAFEF00Dull);  // garbage, like fresh device memory
    std::vector<double> iors_deep((size_t)(kMaxIorsDeep - kMaxIors) * slots, -1.0);  // (garbage: an entry is written before it is read)
    fr.iors_deep = iors_deep.data();
    fr.iors_deep_rows = (uint32_t)(kMaxIorsDeep - kMaxIors);
    for (uint64_t i = 0; i < slots; i++) pool[(size_t)kWfFlags * slots + i] = pool[(size_t)kWfSeq * slots + i] = 0ull;
    const size_t cap = ((size_t)slots + 2 * kWfBlock) * 2;
    std::vector<uint32_t> qwords(2 * cap + 2 * (2 * 8 * cap), 0xA5A5A5A5u);  // item, light, then two sets of eight planes of doubles
    unsigned long long ctrl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, work = 0;
    std::vector<unsigned long long> stats(kStatsWords + 32, 0ull);

    // trace launch (planTrace)
    const uint32_t tblock = trace_waves * 64u;
    std::vector<SmStackEntry> spill((size_t)trace_grid * tblock * d.stack_depth);
    const uint32_t lds_stack = kLdsStackDepth;
    const uint32_t stack_bytes = lds_stack * tblock * (uint32_t)sizeof(SmStackEntry);
    WfTraceArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.stats = stats.data();
    ta.nodes = d.nodes64;
    ta.qblocks = d.qblocks;
    ta.num_nodes = d.q_nodes;
    ta.lds_blocks = (uint32_t)std::min<uint64_t>(d.num_qblocks, ((uint64_t)sizeof(lds) - stack_bytes - 128u - trace_waves * kShareMapBytes) / 64u);
    ta.q_root_a = d.q_root_a;
    ta.q_root_m = d.q_root_m;
    ta.prim = d.prim;
    ta.spill = spill.data();
    ta.total_lanes = trace_grid * tblock;
    ta.refill_lanes = 16;
    ta.leaf_lanes = 16;
    ta.leaf_items = 1 << 20;
    ta.min_inner = 8;
    ta.lds_stack = (int)lds_stack;
    ta.max_stack = d.stack_depth;
    ta.deal_shift = 6;
    ta.pop = ctrl + 2;
    if (trace_form == 0 || trace_form == 1 || trace_form == 2) return -201;  // (forms removed in round 6)
    if (trace_form == 27 && !E.L.q_single) return -204;
    PoolRays pr;
    pr.pool.w = pool.data();
    pr.pool.n = (uint32_t)slots;
    pr.q.item = qwords.data();
    pr.q.light = qwords.data() + cap;
    pr.q.ray = reinterpret_cast<double*>(qwords.data() + 2 * cap);
    pr.q.prev_ray = pr.q.ray + 8 * cap;
    pr.q.cap = cap;
    uint32_t shade_tables = wfShadeTableBytes(d.num_materials, d.num_lights);
    if (shade_tables > kWfShadeTableMax) shade_tables = 0;
    WfShadeArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.pool = pr.pool;
    sa.slot_base = 0;
    sa.slot_count = (uint32_t)slots;
    sa.fr = fr;
    sa.queue = pr.q;
    sa.pop_reset = ctrl + 2;
    sa.work = &work;
    sa.stats = stats.data();
    sa.lds_tables = shade_tables;
    const uint32_t shade_grid = (sa.slot_count + kWfBlock - 1) / kWfBlock;
    // photon mapper: requests and the kNN launch that serves them
    WaveMap wg, wc;
    WfKnnArgs ka;
    memset(&ka, 0, sizeof(ka));
    std::vector<uint32_t> requests, knn_spill;
    std::vector<double> stage, est;
    const uint32_t knn_grid = 2;
    if (photon) {
        if (wg.init(gmap, k_nearest) || wc.init(cmap, k_nearest)) return -301;
        requests.resize(slots);
        stage.resize((size_t)slots * kStageDoubles);
        est.resize((size_t)slots * 6);
        knn_spill.resize((size_t)knn_grid * 4 * kWaveSpill * 3);
        ka.pool = pr.pool;
        ka.requests = requests.data();
        ka.pop = ctrl + 6;
        ka.stats = stats.data();
        ka.maps[0] = wg.view;
        ka.maps[1] = wc.view;
        ka.k = k_nearest;
        ka.stage = stage.data();
        ka.est = est.data();
        ka.spill = knn_spill.data();
        sa.requests = requests.data();
        sa.rpop_reset = ctrl + 6;
        sa.pm.photons[0] = wg.view.base.photons;
        sa.pm.photons[1] = wc.view.base.photons;
        sa.pm.k = k_nearest;
        sa.pm.direct_visualization = direct_visualization != 0;
        sa.pm.est = est.data();
        sa.stage = stage.data();
    }
    uint32_t launches = 0;
    for (uint64_t it = 0;; it++) {
        if (it > 100000) return -400;
        sa.count_out = ctrl + (it & 1);
        sa.count_reset = ctrl + ((it + 1) & 1);
        double* set0 = reinterpret_cast<double*>(qwords.data() + 2 * cap);
        pr.q.ray = set0 + (it & 1) * 8 * cap;
        pr.q.prev_ray = set0 + ((it + 1) & 1) * 8 * cap;
        sa.queue = pr.q;
        if (photon) {
            sa.rcount_out = ctrl + 4 + (it & 1);
            sa.rcount_reset = ctrl + 4 + ((it + 1) & 1);
            launchGrid(shade_grid, kWfBlock, [&] { wfShadeKernel<true>(d, sa); });
        } else {
            launchGrid(shade_grid, kWfBlock, [&] { wfShadeKernel<false>(d, sa); });
        }
        launches++;
        if (ctrl[it & 1] == 0ull && (!photon || ctrl[4 + (it & 1)] == 0ull)) break;  // nothing queued: every slot is done
        ta.count = ctrl + (it & 1);
        for (uint32_t g = 0; g < trace_grid; g++) {
            wemu::launch().block_idx = g;
            wemu::launch().block_dim = tblock;
            wemu::launch().grid_dim = trace_grid;
            wemu::runGroup((int)trace_waves, [&](int) {
                if (trace_form == 11) wfTraceKernel<PoolRays, true, 1>(ta, pr);
                else if (trace_form == 27) wfTraceKernel<PoolRays, true, 3>(ta, pr);
                else wfTraceKernel<PoolRays, true, 0>(ta, pr);
            });
        }
        launches++;
        if (photon) {
            ka.count = ctrl + 4 + (it & 1);
            launchGrid(knn_grid, 256, [&] { wfKnnKernel<true>(ka); });
            launches++;
        }
    }
    launchGrid((uint32_t)((pixels + 255) / 256), 256, [&] { sampleResolveKernel(fr.samples, pixels, fr.spp, out_rgb); });
    launches++;
    if (stats_out) memcpy(stats_out, stats.data(), kStatsWords * sizeof(unsigned long long));
    if (launches_out) *launches_out = launches;
    return stats[5] ? -100 : stats[7] ? -101 : 0;
}

// emitKernel (the photon pass: PhotonMapper's emission loop, photon-mapper.cpp:96-110 / 225-277) on emulated workgroups, its arguments
// filled as emitOnDevice (mcrt_hip.hip) fills them: the work split over the lights, one launch with lists of `capacity` photons, with
// the sizing pilot's stride (1 = every path). Lists out as the device leaves them (unordered); counts[0..1] = photons counted (may
// exceed the capacity: then the lists hold the first `capacity`), counts[2] = paths, counts[3] = rays. Returns 0 / -100 / -101.
int wemu_emit(const mcrt_scene_desc* scene, double emissions, double caustic_factor, uint32_t global_seed, uint32_t stride, uint32_t grid,
              uint64_t capacity, float* out_global, unsigned long long* keys_global, float* out_caustic, unsigned long long* keys_caustic,
              unsigned long long* counts) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    DeviceScene d;
    fillDeviceScene(scene, E, d, 64u);
    d.flat = 0;  // the emission kernel walks the BVH
    const uint32_t nl = scene->num_lights;
    if (nl == 0 || grid == 0 || stride == 0) return -200;
    const size_t photon_emissions = (size_t)((double)(size_t)emissions * caustic_factor);
    double total_add_flux = 0.0;
    std::vector<double> flux((size_t)nl * 3);
    for (uint32_t i = 0; i < nl; i++) {
        const uint32_t ls = scene->light_surface[i];
        for (int c = 0; c < 3; c++) flux[(size_t)i * 3 + c] = scene->materials[scene->surf_material[ls]].emittance[c] * scene->surf_area[ls];
        total_add_flux += 0.0 + flux[(size_t)i * 3] + flux[(size_t)i * 3 + 1] + flux[(size_t)i * 3 + 2];
    }
    std::vector<unsigned long long> first(nl + 1, 0ull);
    std::vector<double> pflux((size_t)nl * 3);
    for (uint32_t i = 0; i < nl; i++) {
        const double* f = &flux[(size_t)i * 3];
        const double share = (0.0 + f[0] + f[1] + f[2]) / total_add_flux;
        const size_t n = (size_t)((double)photon_emissions * share);
        first[i + 1] = first[i] + n;
        for (int c = 0; c < 3; c++) pflux[(size_t)i * 3 + c] = f[c] / (double)n;
    }
    const uint32_t block = kBlock;
    const uint32_t lds_bytes = planLds(d, block).total;
    if (lds_bytes > kEmuMaxLds || lds_bytes > sizeof(lds)) return -202;
    unsigned long long counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<StackEntry> spill((size_t)grid * block * (d.stack_depth - kLdsStackDepth) + 16);
    EmitParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.num_lights = nl;
    prm.light_first = first.data();
    prm.light_photon_flux = pflux.data();
    prm.total_emissions = first[nl];
    prm.first_emission = 0;
    prm.stride = stride;
    prm.global_seed = global_seed;
    prm.non_caustic_reject = 1.0 / caustic_factor;
    prm.photons[0] = out_global;
    prm.photons[1] = out_caustic;
    prm.keys[0] = keys_global;
    prm.keys[1] = keys_caustic;
    prm.capacity[0] = prm.capacity[1] = capacity;
    prm.counters = counters;
    prm.spill = spill.data();
    prm.total_lanes = grid * block;
    if (d.stage_all) launchGrid(grid, block, [&] { emitKernel<true>(d, prm); });
    else launchGrid(grid, block, [&] { emitKernel<false>(d, prm); });
    counts[0] = counters[1];
    counts[1] = counters[2];
    counts[2] = counters[3];
    counts[3] = counters[4];
    return counters[5] ? -100 : counters[6] ? -101 : 0;
}

// 0: the waves of a workgroup take turns; otherwise the seed of a random visiting order (wave_emu.hpp)
void wemu_set_shuffle(unsigned long long seed) { wemu::shuffleSeed() = seed; }

}  // extern "C"
