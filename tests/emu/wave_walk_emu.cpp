// The trace kernels' wave-synchronous tree walk with the shared leaf step (csrc/mcrt_sharedleaf.hpp: traceWalkShared - deferred leaves
// tested by the whole wave, one pop site, the stack's top cached in registers) run on the HOST: 64 rays per emulated wavefront
// (wave_emu.hpp), the device source unchanged. Scene set-up (layout, quantised blocks, stacks) is mcrt_emu.cpp's. Test harness only.
#define MCRT_WAVE_EMU 1
#include "wave_emu.hpp"

#include "mcrt_emu.cpp"

#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_widerec.hpp"

namespace {
inline uint32_t laneId() { return __lane_id(); }
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_sharedleaf.hpp"

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
}  // namespace

extern "C" {

// Closest hits of n rays, 64 per wave; `holes`: every holes-th lane of a wave carries no ray (valid = false), 0 = none.
// which: 0 = traceWalkShared (shared leaf step), 1 = traceWalkQ (every lane its own leaf step). Returns 0, -100 on stack overflow.
int wemu_intersect(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int which, int holes,
                   double* out_t, uint32_t* out_surface, double* out_uv) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    if (scene->num_nodes == 0) return -200;
    QTrace qt;
    qt.init(E, scene);
    const int depth = qt.stk.max_depth;
    std::vector<SmStackEntry> s_lds((size_t)kLdsStackDepth * 64), s_spill((size_t)(depth - kLdsStackDepth) * 64);
    std::vector<uint8_t> map(kShareMapBytes);
    uint32_t overflow = 0;
    // a wave takes the rays a stride apart, so that holes do not drop rays: ray index = wave base + position among the valid lanes
    uint64_t next = 0;
    while (next < n) {
        uint64_t idx[64];
        bool valid[64];
        for (int l = 0; l < 64; l++) {
            valid[l] = !(holes && (l % holes) == 0) && next < n;
            idx[l] = valid[l] ? next++ : 0;
        }
        wemu::run([&](int lane) {
            SmStack stk = qt.stk;
            stk.lds = s_lds.data() + lane;
            stk.lds_stride = 64;
            stk.spill = s_spill.data() + lane;
            stk.spill_stride = 64;
            TraceCounters cnt = {0, 0, 0, 0};
            const uint64_t i = idx[lane];
            Ray ray = makeRay(ld3(start + 3 * i), ld3(direction + 3 * i), 1.0);
            Hit h;
            if (which == 0) {
                h = traceWalkShared<true>(qt.sv, qt.qv, stk, valid[lane], ray, false, nullptr, cnt, map.data());
            } else {
                hitInit(h, kDblMax);
                // (traceWalkQ has no `valid`: lanes without a ray trace a copy of ray 0 and drop the result)
                h = traceWalkQ<true>(qt.sv, qt.qv, stk, ray, false, nullptr, cnt);
            }
            if (valid[lane]) {
                out_t[i] = h.t;
                out_surface[i] = h.surface;
                out_uv[2 * i] = h.u;
                out_uv[2 * i + 1] = h.v;
            }
            if (cnt.overflow) overflow = 1;
        });
    }
    return overflow ? -100 : 0;
}

// The radiance estimates of PhotonMapper::sampleRay at the first hits of n camera rays (pixel i of the frame, sample 0), computed
// twice by the product's code: per lane (estimateCausticRadiance / estimateGlobalRadiance, mcrt_integrator.hpp: the legacy kernel's
// path) and by an emulated wave, 64 hits at a time (stageInteraction -> waveEstimate: search, loadStagedInteraction, waveEvalPhotons
// with its DPP sums - renderKernelPM's part 2). out_lane / out_wave: [n][6] caustic rgb, global rgb; out_valid[i] = 0 where the
// pixel's ray hit nothing or a specular surface (no estimate there). rows: 4 or 16 candidate rows.
int wemu_estimate(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k, int rows,
                  const mcrt_camera_desc* cam, uint32_t global_seed, uint64_t n, double* out_lane, double* out_wave, uint8_t* out_valid) {
    Emu E;
    if (int rc = setup(E, scene, 1)) return rc;
    if (!E.stage_all) return -300;  // (the harness uses the LDS-resident flavour of the views: small scenes)
    setupMap(E, 0, gmap, E.pv.global_map);
    setupMap(E, 1, cmap, E.pv.caustic_map);
    E.pv.k_nearest = k;
    E.pv.direct_visualization = false;
    setupKnn(E, k);
    WaveMap wg, wc;
    if (wg.init(gmap, k) || wc.init(cmap, k)) return -301;
    std::vector<InteractionT<true>> ias(n);
    TraceCounters cnt = {0, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        out_valid[i] = 0;
        for (int c = 0; c < 6; c++) out_lane[6 * i + c] = out_wave[6 * i + c] = 0.0;
        PathState st;
        st.smp.initiate(global_seed, (uint32_t)i);
        st.smp.setIndex(0);
        pathBegin(st, E.rh, cameraRay(*cam, E.sh_all.scene_ior, (uint32_t)(i % cam->width), (uint32_t)(i / cam->width), st.smp, E.tab.data()));
        st.smp.shuffle();
        const Hit isect = sceneIntersect<true, true, false>(E.sv_all, st.ray, E.stk, cnt);
        if (isect.surface == kNoSurface) continue;
        interactionInit(ias[i], E.sh_all, isect, st.ray, E.rh.externalIOR(st.ray), st.smp, E.tab.data());
        if (ias[i].dirac_delta) continue;
        out_valid[i] = 1;
        uint32_t searches = 0, visits = 0;
        const d3 C = estimateCausticRadiance(E.pv, ias[i], E.ks, searches, visits);
        const d3 G = estimateGlobalRadiance(E.pv, ias[i], E.ks, searches, visits);
        out_lane[6 * i + 0] = C.x; out_lane[6 * i + 1] = C.y; out_lane[6 * i + 2] = C.z;
        out_lane[6 * i + 3] = G.x; out_lane[6 * i + 4] = G.y; out_lane[6 * i + 5] = G.z;
    }
    const uint32_t cand = rows == 16 ? waveCand(16) : waveCand(4);
    std::vector<double> d2(cand), stage((size_t)64 * kStageDoubles);
    std::vector<uint32_t> idx(cand), hist(kWaveHist), spill((size_t)3 * kWaveSpill);
    uint32_t overflow_any = 0;
    for (uint64_t base = 0; base < n; base += 64) {
        wemu::run([&](int lane) {
            const uint64_t i = base + (uint64_t)lane;
            const bool want = i < n && out_valid[i];
            WaveKnnLds W;
            W.d2 = d2.data();
            W.idx = idx.data();
            W.hist = hist.data();
            W.spill = spill.data();
            if (want) stageInteraction(stage.data() + (size_t)lane * kStageDoubles, ias[i]);
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();  // (the records are read by the other lanes)
            uint32_t searches = 0, visits = 0, overflow = 0;
            d3 C, G;
            if (rows == 16) {
                C = waveEstimate<true, 16>(want, stage.data(), wc.view, k, true, W, searches, visits, overflow);
                G = waveEstimate<true, 16>(want, stage.data(), wg.view, k, false, W, searches, visits, overflow);
            } else {
                C = waveEstimate<true, 4>(want, stage.data(), wc.view, k, true, W, searches, visits, overflow);
                G = waveEstimate<true, 4>(want, stage.data(), wg.view, k, false, W, searches, visits, overflow);
            }
            if (want) {
                out_wave[6 * i + 0] = C.x; out_wave[6 * i + 1] = C.y; out_wave[6 * i + 2] = C.z;
                out_wave[6 * i + 3] = G.x; out_wave[6 * i + 4] = G.y; out_wave[6 * i + 5] = G.z;
            }
            if (overflow) overflow_any = 1;
        });
    }
    return overflow_any ? -100 : 0;
}

}  // extern "C"
