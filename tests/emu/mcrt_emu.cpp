// tests/emu/mcrt_emu.cpp — TEST HARNESS ONLY (built by tests/conftest.py into tests/emu/_build/).
//
// Compiles the product's per-lane device code (monte-carlo-ray-tracer_amd/csrc/*.hpp: sampler,
// traversal, shading, integrators, kNN — the MCRT_HD functions the gfx950 kernels inline) for the
// HOST and drives it one "lane" at a time, so that the kernel logic can be checked against the
// oracle in the GPU-less container (`pytest -m "not gpu"`). It is not a CPU fallback: nothing in
// the product links or loads it, and libmcrt_hip.so has no host execution path.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_libm.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_libm_pow.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_integrator.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_lanesm.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_layout.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_octree_shared.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_output.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_plan.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_qbvh.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_wavefront.hpp"

using namespace mcrt;

namespace {

struct Emu {
    HostLayout L;
    SceneViewT<true> sv_all;    // "whole scene staged" flavour of the views
    SceneViewT<false> sv_top;   // "top of the BVH staged" flavour
    ShadeViewT<true> sh_all;
    ShadeViewT<false> sh_top;
    bool stage_all;
    std::vector<uint32_t> tab;
    std::vector<StackEntry> stack_lds, stack_spill;
    LaneStack stk;
    std::vector<double> iors;
    RefractionHistory rh;
    std::vector<uint32_t> map_start[2], map_contained[2];
    PhotonViews pv;
    std::vector<double> res_d2, visit_d2;
    std::vector<uint32_t> res_idx, visit_oct;
    KnnScratch ks{};  // (zeroed: path-traced frames never set it up, and report its overflow word with the stacks')
};

int setup(Emu& E, const mcrt_scene_desc* s, int stage_mode) {
    const bool stage_lds = stage_mode != 0;
    std::string err;
    if (int rc = buildLayout(s, E.L, err)) return rc;  // (test harness: the environment is its option channel)
    patchQuadricAddresses(s, E.L, s->quadrics);  // quadric records are read where the descriptor keeps them
    E.tab.resize(kSobolTableWords);
    buildSobolByteTables(E.tab.data());
    E.stack_lds.resize(kLdsStackDepth);
    const int depth = std::max<int>(kMaxStackDepth, (int)E.L.stack_bound + 1);  // as mcrt_upload_scene sizes the device's stacks
    E.stack_spill.resize(depth - kLdsStackDepth);
    E.stk.lds = E.stack_lds.data();
    E.stk.lds_stride = 1;
    E.stk.spill = E.stack_spill.data();
    E.stk.spill_stride = 1;
    E.stk.max_depth = depth;
    E.iors.assign(kMaxIors, 0.0);  // (the wavefront integrator continues deeper histories in the slot's own pool words: wfShadeSlot)
    E.rh.iors = E.iors.data();
    E.rh.stride = 1;
    E.rh.size = 0;
    E.stage_all = stage_lds;
    auto fillScene = [&](auto& sv) {
        sv.num_nodes = s->num_nodes;
        sv.num_surfaces = s->num_surfaces;
        sv.node_bounds = E.L.node_bounds.data();
        sv.node_meta = E.L.node_meta.data();
        sv.prim = E.L.prim.data();
        // the "top" flavour exercises both sides of the node-index test
        sv.lds_nodes = s->num_nodes / 2;
        sv.lds_node_bounds = E.L.node_bounds.data();
        sv.lds_node_meta = E.L.node_meta.data();
        sv.flat_tris = E.L.flat_tris;
        sv.flat_prim = nullptr;
        sv.flat_index = nullptr;
        sv.flat_pre = nullptr;
        sv.pre_tri_pairs = sv.pre_sph_pairs = 0;
        sv.pre_cx = sv.pre_cy = sv.pre_cz = sv.pre_bound = 0.0;
    };
    auto fillShade = [&](auto& sh) {
        sh.surf_v = E.L.num_quadric_surfaces ? E.L.surf_v_patched.data() : s->surf_v;
        sh.surf_normal = E.L.normal.data();
        sh.surf_rec = E.L.shade_rec.data();
        sh.surf_vn = s->surf_vn;
        sh.surf_area = s->surf_area;
        sh.surf_material = s->surf_material;
        sh.surf_kind = s->surf_kind;
        sh.materials = s->materials;
        sh.num_lights = s->num_lights;
        sh.light_surface = s->light_surface;
        sh.light_cdf = s->light_cdf;
        sh.scene_ior = s->scene_ior;
    };
    fillScene(E.sv_all);
    fillScene(E.sv_top);
    fillShade(E.sh_all);
    fillShade(E.sh_top);
    if (stage_mode == 2 || stage_mode == 3) {  // flat (tiny-scene) mode of the "all" flavour; 3: with the FP32 cull in front
        E.sv_all.num_nodes = 0;
        E.sv_all.flat_prim = E.L.flat_prim.data();
        E.sv_all.flat_index = E.L.flat_index.data();
        if (stage_mode == 3) {
            if (E.L.flat_pre.empty()) return MCRT_ERR_UNSUPPORTED;
            E.sv_all.flat_pre = E.L.flat_pre.data();
            E.sv_all.pre_tri_pairs = E.L.pre_tri_pairs;
            E.sv_all.pre_sph_pairs = E.L.pre_sph_pairs;
            E.sv_all.pre_cx = E.L.pre_centre[0];
            E.sv_all.pre_cy = E.L.pre_centre[1];
            E.sv_all.pre_cz = E.L.pre_centre[2];
            E.sv_all.pre_bound = E.L.pre_bound;
        }
    }
    return 0;
}

void setupMap(Emu& E, int which, const mcrt_photon_map_desc* m, PhotonMapView& v) {
    memset(&v, 0, sizeof(v));
    if (!m || m->num_octants == 0) return;
    E.map_start[which].resize(m->num_octants);
    E.map_contained[which].resize(m->num_octants);
    for (uint32_t i = 0; i < m->num_octants; i++) {
        E.map_start[which][i] = (uint32_t)m->octant_start_data[i];
        E.map_contained[which][i] = (uint32_t)m->octant_contained_data[i];
    }
    v.num_octants = m->num_octants;
    v.num_photons = m->num_photons;
    v.octant_bounds = m->octant_bounds;
    v.octant_start = E.map_start[which].data();
    v.octant_contained = E.map_contained[which].data();
    v.octant_next = m->octant_next_sibling;
    v.octant_leaf = m->octant_leaf;
    v.photons = m->photons;
}

void setupKnn(Emu& E, uint32_t k, uint32_t visit_cap = kMaxVisit) {
    E.res_d2.resize(k ? k : 1);
    E.res_idx.resize(k ? k : 1);
    E.visit_d2.resize(visit_cap);
    E.visit_oct.resize(visit_cap);
    E.ks.max_visit = visit_cap;
    E.ks.overflowed = 0u;
    E.ks.res_d2 = E.res_d2.data();
    E.ks.res_idx = E.res_idx.data();
    E.ks.visit_d2 = E.visit_d2.data();
    E.ks.visit_oct = E.visit_oct.data();
    E.ks.stride = 1;
}

}  // namespace

extern "C" {

int emu_render(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
               int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, int integrator, uint32_t row0,
               uint32_t row1, int stage_lds, double* out_rgb, uint64_t* counters /* rays,node_tests,prim_tests,overflow,paths */) {
    Emu E;
    if (int rc = setup(E, scene, stage_lds)) return rc;
    setupMap(E, 0, gmap, E.pv.global_map);
    setupMap(E, 1, cmap, E.pv.caustic_map);
    E.pv.k_nearest = k_nearest;
    E.pv.direct_visualization = direct_visualization != 0;
    setupKnn(E, k_nearest);
    const uint32_t spp = cam->sqrtspp * cam->sqrtspp;
    TraceCounters cnt = {0, 0, 0, 0};
    uint64_t totals[5] = {0, 0, 0, 0, 0};
    uint32_t searches = 0, octant_visits = 0;
    for (uint32_t y = row0; y < row1; y++)
        for (uint32_t x = 0; x < cam->width; x++) {
            PathState st;
            st.smp.initiate(global_seed, y * cam->width + x);
            double acc[3] = {0, 0, 0};
            for (uint32_t i = 0; i < spp; i++) {
                st.smp.setIndex(i);
                pathBegin(st, E.rh, cameraRay(*cam, E.sh_all.scene_ior, x, y, st.smp, E.tab.data()));
                totals[4]++;
                for (;;) {
                    bool done;
                    if (integrator == MCRT_INTEGRATOR_PHOTON_MAPPER)
                        done = E.stage_all ? photonMapperBounce<true, true>(st, E.rh, E.sv_all, E.sh_all, E.pv, E.stk, E.ks, cnt, searches, octant_visits, E.tab.data())
                                           : photonMapperBounce<true, false>(st, E.rh, E.sv_top, E.sh_top, E.pv, E.stk, E.ks, cnt, searches, octant_visits, E.tab.data());
                    else
                        done = stage_lds == 3  ? pathTracerBounce<true, true, false, true>(st, E.rh, E.sv_all, E.sh_all, E.stk, cnt, E.tab.data())  // flat-scene instance
                               : E.stage_all ? pathTracerBounce<true, true>(st, E.rh, E.sv_all, E.sh_all, E.stk, cnt, E.tab.data())
                                             : pathTracerBounce<true, false>(st, E.rh, E.sv_top, E.sh_top, E.stk, cnt, E.tab.data());
                    if (done) break;
                }
                acc[0] += st.radiance.x * 1.0;
                acc[1] += st.radiance.y * 1.0;
                acc[2] += st.radiance.z * 1.0;
                totals[0] += cnt.rays; totals[1] += cnt.node_tests; totals[2] += cnt.prim_tests; totals[3] += cnt.overflow | E.ks.overflowed;
                cnt = TraceCounters{0, 0, 0, 0};
            }
            double* o = out_rgb + ((size_t)(y - row0) * cam->width + x) * 3;
            for (int c = 0; c < 3; c++) o[c] = gmax(acc[c] / (double)spp, 0.0);
        }
    if (counters) memcpy(counters, totals, sizeof(totals));
    return 0;
}

// One ray through the trace kernel's per-lane code: quantised child blocks (mcrt_qbvh.hpp) with the first
// half of the blocks on the "LDS" side, exact records for rays with a zero direction component.
namespace {
int g_defer = 0;
struct QTrace {
    SmSceneView<false> sv;
    QView<true> qv;
    std::vector<SmStackEntry> s_lds, s_spill;
    SmStack stk;
    void init(Emu& E, const mcrt_scene_desc* scene) {
        s_lds.resize(kLdsStackDepth);
        const int depth = std::max<int>(kMaxStackDepth, (int)E.L.stack_bound + 1);
        s_spill.resize(depth - kLdsStackDepth);
        stk.lds = s_lds.data();
        stk.lds_stride = 1;
        stk.spill = s_spill.data();
        stk.spill_stride = 1;
        stk.max_depth = depth;
        sv.num_nodes = (uint32_t)E.L.nodes64.size();  // (a scene without a BVH: the index-range tree of mcrt_layout.hpp)
        sv.nodes = E.L.nodes64.data();
        sv.prim = E.L.prim.data();
        sv.lds_nodes = 0;
        sv.lds_node_ptr = E.L.nodes64.data();
        qv.blocks = E.L.qblocks.data();
        qv.lds_blocks = (uint32_t)E.L.qblocks.size() / 2;
        qv.lds_ptr = E.L.qblocks.data();
        qv.root_a = E.L.q_root_a;
        qv.root_m = E.L.q_root_m;
    }
    // g_defer: 0 = a leaf is tested when it is reached (round 2's order); 1 = deferred leaves (mcrt_lanesm.hpp), the pending leaf
    // tested only when the lane has nothing else to do - the longest a wave's gating can postpone it; k >= 2 = also every k-th
    // iteration (a gate that opens now and then)
    Hit run(d3 o, d3 d, bool shadow, const ShadowQuery* sq, TraceCounters& cnt) {
        Trav T;
        travBeginQ<false, true, true>(sv, qv, T, o, d, rcp3(d), shadow, sq, cnt);
        if (g_defer) {
            PendLeaf P;
            travParkLeaf(T, P, stk);
            for (uint64_t it = 0; T.active || P.n; it++) {
                if (T.active && (T.node_m & kSmInner)) {
                    if (T.fast) travInnerStepQ<true, true>(qv, T, stk, cnt);
                    else travInnerStep<false, true>(sv, T, stk, cnt);
                }
                travParkLeaf(T, P, stk);
                const bool forced = P.n && (!T.active || !(T.node_m & kSmInner));
                if (P.n && (forced || (g_defer >= 2 && it % (uint64_t)g_defer == 0))) travPendStep<false, true>(sv, T, P, cnt);
                travParkLeaf(T, P, stk);
            }
            return T.best;
        }
        while (T.active) {
            if (T.node_m & kSmInner) {
                if (T.fast) travInnerStepQ<true, true>(qv, T, stk, cnt);
                else travInnerStep<false, true>(sv, T, stk, cnt);
            } else {
                travLeafStep<false, true>(sv, T, stk, cnt);
            }
        }
        return T.best;
    }
};
}  // namespace

// FP32 form of the block visit against its FP64 yardstick (mcrt_qbvh.hpp): walks n rays with the FP32 form and, at every
// inner visit, runs the FP64 form on a copy of the state; every child the FP64 form keeps (pushes or continues with) must
// be kept by the FP32 form with a key (entry distance, rounded down) that is not larger. Returns the number of violations;
// out[0] = inner visits, out[1] = children kept by FP64, out[2] = children kept by FP32.
extern "C" int emu_qstep_check(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, uint64_t* out) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return -1;
    if (scene->num_nodes == 0) return -2;
    QTrace qt;
    qt.init(E, scene);
    std::vector<SmStackEntry> a_lds(kMaxStackDepth), b_lds(kMaxStackDepth), dummy(1);
    auto mk = [&](std::vector<SmStackEntry>& v) {
        SmStack s;
        s.lds = v.data();
        s.lds_stride = 1;
        s.spill = dummy.data();
        s.spill_stride = 0;
        s.lds_depth = kMaxStackDepth;
        return s;
    };
    SmStack sa = mk(a_lds), sb = mk(b_lds);
    TraceCounters cnt = {0, 0, 0, 0};
    int bad = 0;
    uint64_t visits = 0, kept64 = 0, kept32 = 0;
    for (uint64_t i = 0; i < n; i++) {
        const d3 o = ld3(start + 3 * i), d = ld3(direction + 3 * i);
        Trav T;
        travBeginQ<false, true, true>(qt.sv, qt.qv, T, o, d, rcp3(d), false, nullptr, cnt);
        while (T.active) {
            if (T.node_m & kSmInner) {
                if (T.fast) {
                    Trav A = T, B = T;
                    A.sp = 0;
                    B.sp = 0;
                    travInnerStepQ64<true, true>(qt.qv, A, sa, cnt);
                    travInnerStepQ<true, true>(qt.qv, B, sb, cnt);
                    // kept sets: pushed entries + the child continued with (if the visit did not end in a pop: sp == 0 and active
                    // after a visit that kept nothing means the state is unchanged/inactive)
                    auto collect = [&](const Trav& X, const std::vector<SmStackEntry>& st, std::vector<std::pair<uint32_t, uint32_t>>& v, bool f32) {
                        for (int k = 0; k < X.sp; k++) v.push_back({st[k].a, st[k].key});
                        (void)f32;
                    };
                    std::vector<std::pair<uint32_t, uint32_t>> v64, v32;
                    collect(A, a_lds, v64, false);
                    collect(B, b_lds, v32, true);
                    // the near child is not on the stack: recover it when the visit kept anything
                    const bool near64 = A.active && (A.sp > 0 || A.node_a != T.node_a || A.node_m != T.node_m);
                    const bool near32 = B.active && (B.sp > 0 || B.node_a != T.node_a || B.node_m != T.node_m);
                    visits++;
                    kept64 += v64.size() + (near64 ? 1 : 0);
                    kept32 += v32.size() + (near32 ? 1 : 0);
                    // a child is identified by its link AND its meta (a leaf's first primitive and an inner child's first block
                    // may be the same number)
                    auto has = [&](uint32_t a, uint32_t m, uint32_t key64) {
                        if (near32 && B.node_a == a && B.node_m == m) return true;  // continued with: nearest, no key to compare
                        for (auto& e : v32)
                            if (e.first == a && (e.second & 0x1FFu) == m) return (e.second & ~0x1FFu) <= (key64 & ~0x1FFu);
                        return false;
                    };
                    for (auto& e : v64)
                        if (!has(e.first, e.second & 0x1FFu, e.second)) {
                            bad++;
                            if (getenv("QDBG")) {
                                fprintf(stderr, "bad: a=%u key64=%08x (t=%g) near32=%d B.node_a=%u; v32:", e.first, e.second, (double)bitsFloat(e.second), (int)near32, B.node_a);
                                for (auto& f : v32) fprintf(stderr, " [%u %08x t=%g]", f.first, f.second, (double)bitsFloat(f.second));
                                fprintf(stderr, " | v64:");
                                for (auto& f : v64) fprintf(stderr, " [%u %08x t=%g]", f.first, f.second, (double)bitsFloat(f.second));
                                fprintf(stderr, " near64=%d A.node_a=%u best=%g\n", (int)near64, A.node_a, T.best.t);
                            }
                        }
                    if (near64 && !has(A.node_a, A.node_m, 0xFFFFFFFFu)) bad++;
                    travInnerStepQ<true, true>(qt.qv, T, qt.stk, cnt);
                } else {
                    travInnerStep<false, true>(qt.sv, T, qt.stk, cnt);
                }
            } else {
                travLeafStep<false, true>(qt.sv, T, qt.stk, cnt);
            }
        }
    }
    out[0] = visits;
    out[1] = kept64;
    out[2] = kept32;
    return bad;
}

// The lane-state-machine integrator (mcrt_lanesm.hpp) driven for one lane at a time: the same
// regenerate / traverse-step / shade / NEE-finish functions the gfx950 kernel calls, without the
// wave-level gating (which only changes how lanes interleave, not what a lane computes).
int emu_render_sm(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t row0, uint32_t row1,
                  int stage_all, double* out_rgb, uint64_t* counters /* rays,node_tests,prim_tests,overflow,paths */) {
    Emu E;
    if (int rc = setup(E, scene, stage_all ? 1 : 0)) return rc;
    const int sm_depth = std::max<int>(kMaxStackDepth, (int)E.L.stack_bound + 1);
    std::vector<SmStackEntry> s_lds(kLdsStackDepth), s_spill(sm_depth - kLdsStackDepth);
    SmStack stk;
    stk.lds = s_lds.data();
    stk.lds_stride = 1;
    stk.spill = s_spill.data();
    stk.spill_stride = 1;
    stk.max_depth = sm_depth;
    SmSceneView<true> sv_all;
    SmSceneView<false> sv_top;
    auto fill = [&](auto& sv) {
        sv.num_nodes = scene->num_nodes;
        sv.nodes = E.L.nodes64.data();
        sv.prim = E.L.prim.data();
        sv.lds_nodes = scene->num_nodes / 2;
        sv.lds_node_ptr = E.L.nodes64.data();
    };
    fill(sv_all);
    fill(sv_top);
    if (scene->num_nodes == 0) return -200;  // the state machine is only used for scenes with a BVH
    const uint32_t spp = cam->sqrtspp * cam->sqrtspp;
    TraceCounters cnt = {0, 0, 0, 0};
    uint64_t paths = 0;
    QTrace qt;  // tree not resident: quantised child blocks, as renderKernelSM<kAll = false>
    qt.init(E, scene);
    auto begin = [&](Trav& T, d3 o, d3 d, d3 inv, bool shadow, const ShadowQuery* sq) {
        if (stage_all) travBegin<true, true>(sv_all, T, o, d, inv, shadow, sq, cnt);
        else travBeginQ<false, true, true>(sv_top, qt.qv, T, o, d, inv, shadow, sq, cnt);
    };
    for (uint32_t y = row0; y < row1; y++)
        for (uint32_t x = 0; x < cam->width; x++) {
            PathState st;
            st.smp.initiate(global_seed, y * cam->width + x);
            double acc[3] = {0, 0, 0};
            for (uint32_t i = 0; i < spp; i++) {
                st.smp.setIndex(i);
                pathBegin(st, E.rh, cameraRay(*cam, E.sh_all.scene_ior, x, y, st.smp, E.tab.data()));
                paths++;
                st.smp.shuffle();
                Trav T;
                NeePending nee;
                nee.pending = false;
                bool alive = false;
                begin(T, st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr);
                for (;;) {
                    while (T.active) {
                        if (T.node_m & kSmInner) {
                            if (stage_all) travInnerStep<true, true>(sv_all, T, stk, cnt);
                            else if (T.fast) travInnerStepQ<true, true>(qt.qv, T, stk, cnt);
                            else travInnerStep<false, true>(sv_top, T, stk, cnt);
                        } else {
                            if (stage_all) travLeafStep<true, true>(sv_all, T, stk, cnt);
                            else travLeafStep<false, true>(sv_top, T, stk, cnt);
                        }
                    }
                    if (T.shadow) {
                        if (stage_all) smNeeFinish(st, E.sh_all, nee, T.best);
                        else smNeeFinish(st, E.sh_top, nee, T.best);
                        nee.pending = false;
                        if (!alive) break;
                        begin(T, st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr);
                        continue;
                    }
                    Ray shadow_ray;
                    ShadowQuery shadow_q;
                    alive = stage_all ? smShade(st, E.rh, E.sh_all, T.best, nee, shadow_ray, shadow_q, E.tab.data())
                                      : smShade(st, E.rh, E.sh_top, T.best, nee, shadow_ray, shadow_q, E.tab.data());
                    if (alive) st.smp.shuffle();
                    if (nee.pending) begin(T, shadow_ray.start, shadow_ray.direction, shadow_ray.inv_direction, true, &shadow_q);
                    else if (alive) begin(T, st.ray.start, st.ray.direction, st.ray.inv_direction, false, nullptr);
                    else break;
                }
                acc[0] += st.radiance.x * 1.0;
                acc[1] += st.radiance.y * 1.0;
                acc[2] += st.radiance.z * 1.0;
            }
            double* o = out_rgb + ((size_t)(y - row0) * cam->width + x) * 3;
            for (int c = 0; c < 3; c++) o[c] = gmax(acc[c] / (double)spp, 0.0);
        }
    if (counters) {
        counters[0] = cnt.rays; counters[1] = cnt.node_tests; counters[2] = cnt.prim_tests; counters[3] = cnt.overflow | E.ks.overflowed; counters[4] = paths;
    }
    return 0;
}


// The wavefront integrator (mcrt_wavefront.hpp): a pool of `slots` path slots in "HBM", and per bounce a shade pass
// over the slots followed by a trace pass over the queued rays — the same wfShadeSlot / wfLoadRay / trav* /
// wfStoreHit the gfx950 kernels run, with the wave-level cooperation (work pop, queue append) done serially.
// The camera's shard fields select the rows, exactly as for mcrt_render.
namespace {
struct HostRayRec {  // one entry of the ray queue (the device keeps these as planes: WfRayQueue)
    uint32_t item, light;
    d3 o, d;
    double dist;
};
struct HostWfEnv {
    unsigned long long* work;
    std::vector<HostRayRec>* queue;
    std::vector<HostRayRec>* prev;  // the queue of the previous shade pass
    void prevRay(uint32_t entry, d3& o, d3& d) const {
        o = (*prev)[entry].o;
        d = (*prev)[entry].d;
    }
    bool any(bool b) const { return b; }
    unsigned long long pop(bool need) const { return need ? (*work)++ : 0ull; }
    void filmAdd(double* a, double v) const { *a += v; }
    std::vector<uint32_t>* requests;
    void request(uint32_t slot, bool want, bool global) const {
        if (want) requests->push_back(slot | (global ? 0x80000000u : 0u));
    }
    void stage(uint32_t, const InteractionT<false>&) const {}  // (the host stand-in of the kNN launch only searches)
    uint32_t push(uint32_t slot, bool p0, bool p1, d3 o0, d3 d0, d3 o1, d3 d1, double dist1, uint32_t light1) const {
        const uint32_t e0 = (uint32_t)queue->size();
        if (p0) queue->push_back(HostRayRec{slot * 2u, kNoSurface, o0, d0, 0.0});
        if (p1) queue->push_back(HostRayRec{slot * 2u + 1u, light1, o1, d1, dist1});
        return e0;
    }
};
}  // namespace

static int renderWf(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                    int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                    double* out_rgb, uint64_t* counters, double* film_out);

int emu_render_wf_pm(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                     int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                     double* out_rgb, uint64_t* counters /* rays,node_tests,prim_tests,overflow,paths,iterations,searches */) {
    return renderWf(scene, gmap, cmap, k_nearest, direct_visualization, cam, global_seed, slots, owned_rows, out_rgb, counters, nullptr);
}

// mcrt_render_film_device / mcrt_film_resolve_device: one shard's splats into a full-frame RGBW buffer; Splat::get over a buffer
int emu_render_wf_film(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                       double* rgbw) {
    if (!filmSplats(cam->film_filter, cam->film_radius)) return -201;
    return renderWf(scene, nullptr, nullptr, 0, 0, cam, global_seed, slots, owned_rows, nullptr, nullptr, rgbw);
}

void emu_film_resolve(const double* rgbw, uint64_t pixels, double* out_rgb) {
    for (uint64_t i = 0; i < pixels; i++) filmResolve(rgbw + i * 4, out_rgb + i * 3);
}

int emu_render_wf(const mcrt_scene_desc* scene, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                  double* out_rgb, uint64_t* counters /* rays,node_tests,prim_tests,overflow,paths,iterations */) {
    return emu_render_wf_pm(scene, nullptr, nullptr, 0, 0, cam, global_seed, slots, owned_rows, out_rgb, counters);
}

// photon = (k_nearest > 0): the photon mapper's wavefront form; the kNN launch is stood in for by the per-lane search of
// mcrt_integrator.hpp (same k photons; the wave-cooperative search itself is device-only code, checked on the GPU)
static int renderWf(const mcrt_scene_desc* scene, const mcrt_photon_map_desc* gmap, const mcrt_photon_map_desc* cmap, uint32_t k_nearest,
                    int direct_visualization, const mcrt_camera_desc* cam, uint32_t global_seed, uint32_t slots, uint32_t owned_rows,
                    double* out_rgb, uint64_t* counters, double* film_out) {
    const bool photon = k_nearest > 0;
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    PhotonMapView maps[2];
    std::vector<uint32_t> res_n, res_idx;
    std::vector<double> res_r2, res_d2;
    WfPmView pm;
    memset(&pm, 0, sizeof(pm));
    if (photon) {
        setupMap(E, 0, gmap, maps[0]);
        setupMap(E, 1, cmap, maps[1]);
        setupKnn(E, k_nearest);
        res_n.assign((size_t)2 * slots, 0);
        res_r2.assign((size_t)2 * slots, 0.0);
        res_idx.assign((size_t)2 * k_nearest * slots, 0);
        res_d2.assign((size_t)2 * k_nearest * slots, 0.0);
        pm.photons[0] = maps[0].photons;
        pm.photons[1] = maps[1].photons;
        pm.res_n = res_n.data();
        pm.res_r2 = res_r2.data();
        pm.res_idx = res_idx.data();
        pm.res_d2 = res_d2.data();
        pm.k = k_nearest;
        pm.direct_visualization = direct_visualization != 0;
    }
    if (slots == 0 || E.L.nodes64.empty()) return -200;
    QTrace qt;
    qt.init(E, scene);

    // as on the device, where only the flags and the kWfSeq planes of a fresh pool are cleared: every other word starts as garbage
    // (all ones = NaN as a double, 4e9 as an index), so a word that is read before it is written shows in the frame
    std::vector<unsigned long long> pool((size_t)kWfWords * slots, ~0ull);
    std::vector<double> iors_deep((size_t)(kMaxIorsDeep - kMaxIors) * slots, -1.0);  // (garbage: an entry is written before it is read)
    for (uint32_t i = 0; i < slots; i++) pool[(size_t)kWfFlags * slots + i] = pool[(size_t)kWfSeq * slots + i] = 0ull;
    WfPool P;
    P.w = pool.data();
    P.n = slots;
    WfFrame fr;
    fr.cam = *cam;
    fr.global_seed = global_seed;
    fr.spp = cam->sqrtspp * cam->sqrtspp;
    fr.tiles_x = (cam->width + 7) / 8;
    fr.iors_deep = iors_deep.data();
    fr.iors_deep_rows = (uint32_t)(kMaxIorsDeep - kMaxIors);
    // one pass over the owned rows; units per pixel 1, 2 or 4 depending on the slot count, so that the tests cover whole-pixel
    // units, chunks, and chunk counts that do not divide spp (empty last chunk)
    fr.chunk_shift = slots % 3u;
    fr.chunk = (fr.spp + (1u << fr.chunk_shift) - 1u) >> fr.chunk_shift;
    fr.row_base = 0;
    fr.row_end = owned_rows;
    fr.pass_pixels = (unsigned long long)owned_rows * cam->width;
    fr.work_items = ((unsigned long long)fr.tiles_x * ((owned_rows + 7) / 8) * 64ull) << fr.chunk_shift;
    std::vector<double> samples;
    fr.film.type = MCRT_FILM_BOX;
    std::vector<double> blob, cache;
    if (filmSplats(cam->film_filter, cam->film_radius)) {  // as launchWavefront sets the film up
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
            cache.resize(f.cache_size);
            for (uint32_t i = 0; i < f.cache_size; i++) cache[i] = filmFilterFunction(f.type, (2.0 * (int)i) / (double)(f.cache_size - 1));
            f.cache = cache.data();
            f.inv_dx = (double)(f.cache_size - 1) / f.radius;
        }
        if (film_out) std::fill(film_out, film_out + (size_t)cam->width * cam->height * 4, 0.0);
        else blob.assign((size_t)cam->width * cam->height * 4, 0.0);
        f.blob = film_out ? film_out : blob.data();
    }
    if (fr.film.type == MCRT_FILM_BOX) {
        samples.assign((size_t)fr.spp * fr.pass_pixels * 3, 0.0);
        fr.samples = samples.data();
    }
    unsigned long long work = 0;
    std::vector<HostRayRec> queue, prev_queue;
    std::vector<uint32_t> requests;
    HostWfEnv env{&work, &queue, &prev_queue, &requests};
    TraceCounters cnt = {0, 0, 0, 0};
    uint32_t paths = 0;
    uint64_t iterations = 0, searches = 0;
    for (;;) {
        queue.swap(prev_queue);
        queue.clear();
        requests.clear();
        for (uint32_t s = 0; s < slots; s++) {
            if (photon) wfShadeSlot<false, true>(env, P, s, wfSlotFlags(P, s, true), fr, E.sh_top, E.rh, E.tab.data(), paths, &pm);
            else wfShadeSlot<false, false>(env, P, s, wfSlotFlags(P, s, true), fr, E.sh_top, E.rh, E.tab.data(), paths);
        }
        iterations++;
        if (queue.empty() && requests.empty()) break;
        for (uint32_t req : requests) {  // the kNN launch
            const uint32_t slot = req & 0x7FFFFFFFu;
            const d3 p = P.get3(kWfRayO, slot) + P.get3(kWfRayD, slot) * P.getd(kWfHit0T, slot);
            for (int map = 1; map >= ((req >> 31) ? 0 : 1); map--) {
                uint32_t visits = 0;
                const uint32_t c = knnSearch(maps[map], p, k_nearest, E.ks, visits);
                searches++;
                res_n[(size_t)map * slots + slot] = c;
                res_r2[(size_t)map * slots + slot] = c ? E.ks.res(0).distance2 : 0.0;  // heap top = the farthest of the k
                for (uint32_t j = 0; j < c; j++) {
                    const size_t at = ((size_t)map * k_nearest + j) * slots + slot;
                    res_idx[at] = E.ks.res(j).index;
                    res_d2[at] = E.ks.res(j).distance2;
                }
            }
        }
        for (const HostRayRec& r : queue) {
            ShadowQuery sq;
            sq.t_near = 0.0;
            sq.t_far = kDblMax;
            if (r.item & 1u) sq.setRange(r.dist);
            sq.light = r.light;
            const Hit h = qt.run(r.o, r.d, (r.item & 1u) != 0u, &sq, cnt);
            wfStoreHit(P, r.item, h);
        }
    }
    if (fr.film.type != MCRT_FILM_BOX && !film_out)
        for (size_t i = 0; i < (size_t)cam->width * cam->height; i++) filmResolve(&blob[i * 4], out_rgb + i * 3);
    if (fr.film.type == MCRT_FILM_BOX)  // sampleResolveKernel: a pixel's samples added in sample order, Splat::get
        for (size_t i = 0; i < (size_t)fr.pass_pixels; i++) {
            double acc[3] = {0.0, 0.0, 0.0};
            for (uint32_t sidx = 0; sidx < fr.spp; sidx++)
                for (int c = 0; c < 3; c++) acc[c] += samples[((size_t)sidx * fr.pass_pixels + i) * 3 + c];
            for (int c = 0; c < 3; c++) out_rgb[i * 3 + c] = gmax(acc[c] / (double)fr.spp, 0.0);
        }
    if (counters) {
        counters[0] = cnt.rays; counters[1] = cnt.node_tests; counters[2] = cnt.prim_tests; counters[3] = cnt.overflow | E.ks.overflowed; counters[4] = paths;
        counters[5] = iterations;
        if (photon) counters[6] = searches;
    }
    return 0;
}

// Photon emission pass (emitBegin / emitBounce, the functions emitKernel runs), one photon path at a time.
int emu_emit_photons(const mcrt_scene_desc* scene, double emissions, double caustic_factor, uint32_t global_seed, int stage_all,
                     float* gph, uint64_t* gkeys, uint64_t gcap, uint64_t* gcount, float* cph, uint64_t* ckeys, uint64_t ccap,
                     uint64_t* ccount, uint64_t* rays) {
    Emu E;
    if (int rc = setup(E, scene, stage_all ? 1 : 0)) return rc;
    const size_t photon_emissions = (size_t)((double)(size_t)emissions * caustic_factor);
    const double non_caustic_reject = 1.0 / caustic_factor;
    std::vector<double> lf((size_t)scene->num_lights * 3);
    double total = 0.0;
    for (uint32_t i = 0; i < scene->num_lights; i++) {
        const uint32_t ls = scene->light_surface[i];
        for (int c = 0; c < 3; c++) lf[i * 3 + c] = scene->materials[scene->surf_material[ls]].emittance[c] * scene->surf_area[ls];
        total += 0.0 + lf[i * 3] + lf[i * 3 + 1] + lf[i * 3 + 2];
    }
    TraceCounters cnt = {0, 0, 0, 0};
    uint64_t ng = 0, nc = 0;
    for (uint32_t i = 0; i < scene->num_lights; i++) {
        const double share = (0.0 + lf[i * 3] + lf[i * 3 + 1] + lf[i * 3 + 2]) / total;
        const size_t n = (size_t)((double)photon_emissions * share);
        const d3 pf = d3{lf[i * 3] / (double)n, lf[i * 3 + 1] / (double)n, lf[i * 3 + 2] / (double)n};
        for (size_t j = 0; j < n; j++) {
            EmitState es;
            if (stage_all) emitBegin(es, E.rh, E.sh_all, i, (uint32_t)j, pf, global_seed, E.tab.data());
            else emitBegin(es, E.rh, E.sh_top, i, (uint32_t)j, pf, global_seed, E.tab.data());
            for (;;) {
                PhotonOut out;
                const bool done = stage_all ? emitBounce<true, true>(es, E.rh, E.sv_all, E.sh_all, E.stk, cnt, E.tab.data(), non_caustic_reject, out)
                                            : emitBounce<true, false>(es, E.rh, E.sv_top, E.sh_top, E.stk, cnt, E.tab.data(), non_caustic_reject, out);
                if (out.store) {
                    uint64_t& n_out = out.caustic ? nc : ng;
                    const uint64_t cap = out.caustic ? ccap : gcap;
                    if (n_out < cap) {
                        memcpy((out.caustic ? cph : gph) + n_out * 8, out.rec, 32);
                        (out.caustic ? ckeys : gkeys)[n_out] = out.key;
                    }
                    n_out++;
                }
                if (done) break;
            }
        }
    }
    *gcount = ng;
    *ccount = nc;
    *rays = cnt.rays;
    return (ng > gcap || nc > ccap) ? -1 : (cnt.overflow ? -100 : 0);
}

// Leaf-deferral policy of the trace kernel's per-lane code in this emulation (QTrace::run) and its test counts.
void emu_set_defer(int policy) { g_defer = policy; }
// refSinCos / refSin / refCos (mcrt_libm.hpp) against this host's libm on n arguments: uniform in [lo, hi] (splitmix64 stream),
// then - when edges != 0 - the same count again clustered within a few thousand ulps of the branch boundaries of s_sin.c. The
// three libm functions are called through function pointers, one per call: a sin and a cos of the same argument in one
// expression would be merged into sincos by the compiler. out[0..3] = arguments on which sincos' sine / sincos' cosine / sin /
// cos differ (bitwise) from the restatement; out[4..7] = the first such argument of each (bit pattern), if any.
void emu_libm_check(uint64_t n, uint64_t seed, double lo, double hi, int edges, uint64_t* out) {
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    double (*volatile libm_sin)(double) = ::sin;
    double (*volatile libm_cos)(double) = ::cos;
    void (*volatile libm_sincos)(double, double*, double*) = ::sincos;
    for (int i = 0; i < 8; i++) out[i] = 0;
    auto note = [&](int w, double got, double want, double x) {
        if (memcmp(&got, &want, 8) != 0) {
            if (!out[w]) memcpy(&out[4 + w], &x, 8);
            out[w]++;
        }
    };
    auto check = [&](double x) {
        double s2, c2, rs, rc;
        libm_sincos(x, &s2, &c2);
        refSinCos(x, rs, rc);
        note(0, rs, s2, x);
        note(1, rc, c2, x);
        note(2, refSin(x), libm_sin(x), x);
        note(3, refCos(x), libm_cos(x), x);
    };
    for (uint64_t i = 0; i < n; i++) check(lo + (hi - lo) * ((double)(next() >> 11) * 0x1.0p-53));
    if (edges) {
        const double marks[] = {0x1.0p-26, 0x1.0p-27, 0.126, 0.855469, 2.426265, 0.78539816339744831, 1.5707963267948966, 3.1415926535897931,
                                4.7123889803846897, 6.2831853071795862, 105414350.0, 0.0078125, 0.5, 1.0, 2.0};
        for (uint64_t i = 0; i < n; i++) {
            double m = marks[next() % (sizeof(marks) / sizeof(marks[0]))];
            long long b;
            memcpy(&b, &m, 8);
            b += (long long)(next() % 8192) - 4096;
            memcpy(&m, &b, 8);
            check((next() & 1) ? m : -m);
        }
    }
}

// refAsin (mcrt_libm.hpp) against this host's asin: n arguments uniform in [lo, hi], then - edges != 0 - n more within a few thousand
// ulps of every interval boundary of e_asin.c (multiples of 2^-8 in [0.125, 1], 2^-26, 1). out[0] = arguments on which the FMA-form
// restatement differs (bitwise) from libm, out[1] = the first of them (bit pattern); out[2], out[3]: the same for the all-operations-
// rounded form (kFused = false; informational: libm's IFUNC picks ONE variant per machine).
void emu_asin_check(uint64_t n, uint64_t seed, double lo, double hi, int edges, uint64_t* out) {
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    double (*volatile libm_asin)(double) = ::asin;
    for (int i = 0; i < 4; i++) out[i] = 0;
    auto check = [&](double x) {
        const double want = libm_asin(x), a = refAsinT<true>(x), b = refAsinT<false>(x);
        const bool nan_ok = want != want;
        if (nan_ok ? (a == a) : memcmp(&a, &want, 8) != 0) {
            if (!out[0]) memcpy(&out[1], &x, 8);
            out[0]++;
        }
        if (nan_ok ? (b == b) : memcmp(&b, &want, 8) != 0) {
            if (!out[2]) memcpy(&out[3], &x, 8);
            out[2]++;
        }
    };
    for (uint64_t i = 0; i < n; i++) check(lo + (hi - lo) * ((double)(next() >> 11) * 0x1.0p-53));
    if (edges) {
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t r = next();
            double m = (r % 227u) == 0u ? 0x1.0p-26 : (double)(32u + (r % 225u)) / 256.0;  // 0.125 .. 1.0 in steps of 2^-8
            long long b;
            memcpy(&b, &m, 8);
            b += (long long)(next() % 8192) - 4096;
            memcpy(&m, &b, 8);
            check((next() & 1) ? m : -m);
        }
    }
}

// refAtan2 (mcrt_libm.hpp) against this host's atan2 on n argument pairs per family: (0) directions - y = r sin a, x = r cos a,
// a uniform in (-pi, pi], r log-uniform over [2^-scale, 2^scale]; (1) components of unit vectors as the photon constructor passes them
// (length of xy against z; y against x); (2) ratios at the table-interval boundaries (u = k / 512 +- a few thousand ulps, both
// orders, all sign combinations); (3) extreme ratios (2^+-40 .. 2^+-70), zeros, axes, subnormals. out[0] = pairs on which the
// restatement differs (bitwise) from libm, out[1], out[2] = the first such y, x (bit patterns).
void emu_atan2_check(uint64_t n, uint64_t seed, int scale, uint64_t* out) {
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto unit = [&]() { return (double)(next() >> 11) * 0x1.0p-53; };
    double (*volatile libm_atan2)(double, double) = ::atan2;
    out[0] = out[1] = out[2] = 0;
    auto check = [&](double y, double x) {
        const double want = libm_atan2(y, x), got = refAtan2(y, x);
        if (want != want ? (got == got) : memcmp(&got, &want, 8) != 0) {
            if (!out[0]) {
                memcpy(&out[1], &y, 8);
                memcpy(&out[2], &x, 8);
            }
            out[0]++;
        }
    };
    for (uint64_t i = 0; i < n; i++) {  // (0)
        const double a = (2.0 * unit() - 1.0) * 3.14159265358979323846, r = ldexp(1.0 + unit(), (int)(next() % (uint64_t)(2 * scale + 1)) - scale);
        check(r * ::sin(a), r * ::cos(a));
    }
    for (uint64_t i = 0; i < n; i++) {  // (1)
        double d[3] = {2.0 * unit() - 1.0, 2.0 * unit() - 1.0, 2.0 * unit() - 1.0};
        const double l = ::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(l > 1e-3)) continue;
        for (int k = 0; k < 3; k++) d[k] /= l;
        check(::sqrt(d[0] * d[0] + d[1] * d[1]), d[2]);
        check(d[1], d[0]);
    }
    for (uint64_t i = 0; i < n; i++) {  // (2)
        double u = (double)(1 + next() % 512u) / 512.0;
        long long b;
        memcpy(&b, &u, 8);
        b += (long long)(next() % 8192) - 4096;
        memcpy(&u, &b, 8);
        const double big = ldexp(1.0 + unit(), (int)(next() % 41u) - 20), small = big * u;
        const uint64_t r = next();
        const double p = (r & 1) ? small : big, q = (r & 1) ? big : small;
        check((r & 2) ? -p : p, (r & 4) ? -q : q);
    }
    for (uint64_t i = 0; i < n / 4 + 64; i++) {  // (3)
        const uint64_t r = next();
        const double big = ldexp(1.0 + unit(), (int)(r % 61u) - 30), ratio = ldexp(1.0 + unit(), -(int)(40 + (r >> 8) % 31u));
        double p = big, q = big * ratio;
        switch ((r >> 16) % 8u) {
            case 0: q = 0.0; break;
            case 1: q = -0.0; break;
            case 2: p = 0x1.0p-1060 * (1.0 + (double)((r >> 20) % 1000u)); break;
            case 3: p = ldexp(1.0 + unit(), 600); q = ldexp(1.0 + unit(), 590 + (int)((r >> 20) % 21u)); break;
            case 4: p = ldexp(1.0 + unit(), -600); q = ldexp(1.0 + unit(), -590 - (int)((r >> 20) % 21u)); break;
            default: break;
        }
        const bool sw = (r >> 40) & 1;
        const double yy = sw ? q : p, xx = sw ? p : q;
        check(((r >> 41) & 1) ? -yy : yy, ((r >> 42) & 1) ? -xx : xx);
    }
}

// refSinCosF (mcrt_libm.hpp) against this host's sincosf, sinf and cosf - each through its own volatile pointer - on EVERY float whose
// bit pattern lies in [first, last] (both signs: the pattern with and without the sign bit). out: {arguments where the sine / the cosine
// of sincosf differs, where sinf / cosf differ from the restatement, first differing argument's bits}.
void emu_sincosf_check(uint32_t first, uint32_t last, uint64_t* out) {
    float (*volatile f_sin)(float) = ::sinf;
    float (*volatile f_cos)(float) = ::cosf;
    void (*volatile f_sincos)(float, float*, float*) = ::sincosf;
    out[0] = out[1] = out[2] = out[3] = out[4] = 0;
    for (uint64_t b = first; b <= last; b++)
        for (uint32_t sgn = 0; sgn < 2; sgn++) {
            const uint32_t bits = (uint32_t)b | (sgn << 31);
            const float y = bitsFloat(bits);
            float s0, c0, s1, c1;
            refSinCosF(y, s0, c0);
            f_sincos(y, &s1, &c1);
            const float s2 = f_sin(y), c2 = f_cos(y);
            const bool bad_s = floatBits(s0) != floatBits(s1), bad_c = floatBits(c0) != floatBits(c1);
            const bool bad_s2 = floatBits(s0) != floatBits(s2), bad_c2 = floatBits(c0) != floatBits(c2);
            if ((bad_s || bad_c || bad_s2 || bad_c2) && !(out[0] | out[1] | out[2] | out[3])) out[4] = bits;
            out[0] += bad_s;
            out[1] += bad_c;
            out[2] += bad_s2;
            out[3] += bad_c2;
        }
}

// refPow (mcrt_libm_pow.hpp) against this host's pow. family 0: n arguments x uniform in [lo, hi] with y = 1 / 2.4 (sRGB::gammaCompress);
// family 1: x = 2^e m with e uniform in [-300, 300], m in [1, 2), y uniform in [-2.4, 2.4]; family 2: x within a few thousand ulps of 1 and
// of the table's interval boundaries (bits OFF + (i << 45)), y = 1 / 2.4 and y uniform in [-2.4, 2.4] alternately (|y log x| < 512 throughout:
// the restated main path; beyond it refPow IS the platform's pow). out: {differing results,
// bits of the first differing x, bits of its y}.
void emu_pow_check(int family, uint64_t n, double lo, double hi, uint64_t seed, uint64_t* out) {
    double (*volatile f_pow)(double, double) = ::pow;
    out[0] = out[1] = out[2] = 0;
    uint64_t state = seed;
    auto next = [&]() {
        uint64_t zz = (state += 0x9E3779B97F4A7C15ull);
        zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
        zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
        return zz ^ (zz >> 31);
    };
    auto unit = [&]() { return (double)(next() >> 11) * 0x1.0p-53; };
    const double g = 1.0 / 2.4;
    for (uint64_t i = 0; i < n; i++) {
        double x, y;
        if (family == 0) {
            x = lo + (hi - lo) * unit();
            y = g;
        } else if (family == 1) {
            const int e = (int)(next() % 601u) - 300;
            x = ldexp(1.0 + unit(), e);
            y = -2.4 + 4.8 * unit();
        } else {
            const uint64_t r = next();
            const uint64_t centre = (r & 1u) ? 0x3ff0000000000000ull : 0x3fe6955500000000ull + (((r >> 1) & 127u) << 45);
            x = bitsD(centre + ((r >> 8) & 0x1FFFu) - 0x1000u);
            y = (r & 0x200000u) ? g : -2.4 + 4.8 * unit();
        }
        const double a = refPow(x, y), b = f_pow(x, y);
        if (dBits(a) != dBits(b)) {
            if (!out[0]) {
                out[1] = dBits(x);
                out[2] = dBits(y);
            }
            out[0]++;
        }
    }
}

// This host's libm on arrays (the expected values of the GPU known-answer test of mcrt_libm): fn as MCRT_LIBM_*. Every function is
// called through its own volatile pointer, one call per argument (a sin and a cos of one argument would be merged into sincos).
void emu_libm_host(int fn, uint64_t n, const double* a, const double* b, double* out0, double* out1) {
    double (*volatile f_sin)(double) = ::sin;
    double (*volatile f_cos)(double) = ::cos;
    double (*volatile f_asin)(double) = ::asin;
    double (*volatile f_atan2)(double, double) = ::atan2;
    void (*volatile f_sincos)(double, double*, double*) = ::sincos;
    void (*volatile f_sincosf)(float, float*, float*) = ::sincosf;
    double (*volatile f_pow)(double, double) = ::pow;
    for (uint64_t i = 0; i < n; i++) {
        if (fn == 6) {
            out0[i] = f_pow(a[i], b[i]);
        } else if (fn == 5) {
            float sn, cs;
            f_sincosf((float)a[i], &sn, &cs);
            out0[i] = (double)sn;
            out1[i] = (double)cs;
        } else if (fn == 0) f_sincos(a[i], &out0[i], &out1[i]);
        else if (fn == 1) out0[i] = f_sin(a[i]);
        else if (fn == 2) out0[i] = f_cos(a[i]);
        else if (fn == 3) out0[i] = f_asin(a[i]);
        else out0[i] = f_atan2(a[i], b[i]);
    }
}

int emu_trace_counts(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int policy, uint64_t* out /* node tests, primitive tests */) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    if (scene->num_nodes == 0) return -200;
    QTrace qt;
    qt.init(E, scene);
    TraceCounters cnt = {0, 0, 0, 0};
    const int keep = g_defer;
    g_defer = policy < 0 ? 0 : policy;
    for (uint64_t i = 0; i < n; i++) qt.run(ld3(start + 3 * i), ld3(direction + 3 * i), false, nullptr, cnt);
    g_defer = keep;
    out[0] = cnt.node_tests;
    out[1] = cnt.prim_tests;
    return cnt.overflow ? -100 : 0;
}

int emu_intersect(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int stage_lds,
                  double* out_t, uint32_t* out_surface, double* out_uv) {
    Emu E;
    if (int rc = setup(E, scene, stage_lds == 3 ? 0 : stage_lds == 4 ? 3 : stage_lds)) return rc;  // 4: flat loop behind the FP32 cull
    TraceCounters cnt = {0, 0, 0, 0};
    QTrace qt;
    if (stage_lds == 3) {
        if (scene->num_nodes == 0) return -200;
        qt.init(E, scene);
    }
    for (uint64_t i = 0; i < n; i++) {
        Ray ray = makeRay(ld3(start + 3 * i), ld3(direction + 3 * i), 1.0);
        Hit h = stage_lds == 3 ? qt.run(ray.start, ray.direction, false, nullptr, cnt)
                : stage_lds == 4 ? sceneIntersect<true, true, false, true>(E.sv_all, ray, E.stk, cnt)
                : E.stage_all ? sceneIntersect<true, true, false>(E.sv_all, ray, E.stk, cnt)
                            : sceneIntersect<false, true, false>(E.sv_top, ray, E.stk, cnt);
        out_t[i] = h.t;
        out_surface[i] = h.surface;
        out_uv[2 * i] = h.u;
        out_uv[2 * i + 1] = h.v;
    }
    return cnt.overflow ? -100 : 0;
}

// FP32 cull of the flat loop (mcrt_scene.hpp): per ray, the survivor masks and the primitives the FP64 tests accept.
// out[4 i + 0..3] = the numbers of {triangle survivors, sphere survivors, accepted triangles, accepted spheres};
// the return value counts accepted primitives that the cull dropped (must be 0).
int emu_flat_cull(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, uint32_t* out) {
    Emu E;
    if (int rc = setup(E, scene, 3)) return rc;
    const SceneViewT<true>& sv = E.sv_all;
    const uint32_t nt = sv.flat_tris, ns = sv.num_surfaces;
    int missed = 0;
    for (uint64_t i = 0; i < n; i++) {
        const d3 o = ld3(start + 3 * i), d = ld3(direction + 3 * i);
        const CullRay cr = cullRay(sv, o, d);
        uint64_t mt = 0, ms = 0, at = 0, as = 0;  // (<= 64 primitives per kind: MCRT_FLAT_MAX)
        for (uint32_t base = 0; base < nt; base += 32u)
            mt |= (uint64_t)cullTriangles(sv.flat_pre + (size_t)(base / 2u) * kTriPairFloats, std::min(16u, sv.pre_tri_pairs - base / 2u),
                                          std::min(32u, nt - base), cr) << base;
        for (uint32_t base = 0; base < ns - nt; base += 32u)
            ms |= (uint64_t)cullSpheres(sv.flat_pre + (size_t)sv.pre_tri_pairs * kTriPairFloats + (size_t)(base / 2u) * kSphPairFloats,
                                        std::min(16u, sv.pre_sph_pairs - base / 2u), std::min(32u, ns - nt - base), cr) << base;
        for (uint32_t j = 0; j < nt; j++) {
            double t, u, v;
            if (triangleTestFlat(sv.flat_prim + (size_t)j * kPrimStride, o, d, t, u, v)) at |= 1ull << j;
        }
        for (uint32_t j = nt; j < ns; j++) {
            double t;
            if (sphereTestFlat(sv.flat_prim + (size_t)j * kPrimStride, o, d, t)) as |= 1ull << (j - nt);
        }
        missed += __builtin_popcountll(at & ~mt) + __builtin_popcountll(as & ~ms);
        out[4 * i + 0] = (uint32_t)__builtin_popcountll(mt);
        out[4 * i + 1] = (uint32_t)__builtin_popcountll(ms);
        out[4 * i + 2] = (uint32_t)__builtin_popcountll(at);
        out[4 * i + 3] = (uint32_t)__builtin_popcountll(as);
    }
    return missed;
}

// mcrt_photon_map_build_gpu with its device steps done on the host: the same per-photon cell code
// (photonCellCode), a stable sort standing in for the radix sort, the same assembler, the leaf boxes by a plain loop.
// Returns 1 when the builder would fall back to the recursive host builder (cell deeper than the codes).
int emu_octree_build(const float* photons, uint64_t n, const double* bb_min, const double* bb_max, uint32_t max_per_leaf,
                     mcrt_photon_map** out) {
    mcrt_photon_map* M = new mcrt_photon_map();
    if (n == 0) {
        finishMapDesc(M);
        *out = M;
        return 0;
    }
    std::vector<unsigned long long> code(n);
    std::vector<uint32_t> index(n);
    for (uint64_t i = 0; i < n; i++) {
        code[i] = photonCellCode(photons + i * 8, bb_min, bb_max);
        index[i] = (uint32_t)i;
    }
    std::stable_sort(index.begin(), index.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    std::vector<unsigned long long> keys(n);
    M->photons.resize(n * 8);
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = code[index[i]];
        memcpy(&M->photons[i * 8], photons + (size_t)index[i] * 8, 32);
    }
    OctreeAssembler A;
    A.keys = keys.data();
    A.max_node_data = max_per_leaf;
    A.M = M;
    A.node(0, n, 0, true, 0xFFFFFFFFu);
    M->bounds.assign(M->start.size() * 6, 0.0);
    for (uint32_t l : A.leaves) {
        double* bb = &M->bounds[(size_t)l * 6];
        for (int c = 0; c < 3; c++) {
            bb[c] = 1.7976931348623157e308;
            bb[3 + c] = -1.7976931348623157e308;
        }
        for (uint64_t i = M->start[l]; i < M->start[l] + M->contained[l]; i++)
            for (int c = 0; c < 3; c++) {
                const double v = (double)M->photons[i * 8 + 3 + c];
                if (bb[c] > v) bb[c] = v;
                if (bb[3 + c] < v) bb[3 + c] = v;
            }
    }
    A.mergeBounds();
    finishMapDesc(M);
    *out = M;
    return A.too_deep ? 1 : 0;
}

const mcrt_photon_map_desc* emu_octree_desc(const mcrt_photon_map* m) { return &m->desc; }
void emu_octree_free(mcrt_photon_map* m) { delete m; }

void emu_sampler(uint32_t global_seed, uint32_t pixel, uint32_t index, uint32_t shuffles, double* out) {
    static std::vector<uint32_t> tab;
    if (tab.empty()) {
        tab.resize(kSobolTableWords);
        buildSobolByteTables(tab.data());
    }
    Sampler s;
    s.initiate(global_seed, pixel);
    s.setIndex(index);
    for (uint32_t i = 0; i < shuffles; i++) s.shuffle();
    for (int d = 0; d < 7; d++) out[d] = s.get(d, tab.data());
}

// The same numbers from Sampler::restore — what the wavefront pool keeps of a sampler is (pixel, sample index, number of shuffles)
void emu_sampler_restore(uint32_t global_seed, uint32_t pixel, uint32_t index, uint32_t shuffles, double* out) {
    static std::vector<uint32_t> tab;
    if (tab.empty()) {
        tab.resize(kSobolTableWords);
        buildSobolByteTables(tab.data());
    }
    Sampler s;
    s.base_seed = s.seed = s.sequence = s.bit_reversed_index = s.shuffled_index = 0xDEADBEEFu;
    s.restore(global_seed, pixel, index, shuffles);
    for (int d = 0; d < 7; d++) out[d] = s.get(d, tab.data());
    out[7] = (double)s.sequence;
}

// HostLayout::shade_rec of a scene: [num_surfaces][16] doubles (the bits of word 3 = material | kind << 32)
int emu_shade_rec(const mcrt_scene_desc* scene, double* out) {
    HostLayout L;
    std::string err;
    if (int rc = buildLayout(scene, L, err)) return rc;
    memcpy(out, L.shade_rec.data(), L.shade_rec.size() * sizeof(double));
    return 0;
}

// visit_cap: the per-lane frontier's capacity (0: the library's default, 160). Returns 0, or 1 when a search ran out of frontier (the
// host of the library then repeats the work with eight times the capacity; here the caller does).
int emu_knn_cap(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k, uint32_t visit_cap, uint32_t* out_count, uint32_t* out_index,
                double* out_d2);
int emu_knn(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k, uint32_t* out_count, uint32_t* out_index,
            double* out_d2) {
    for (uint32_t cap = kMaxVisit;; cap *= 8u) {  // (as mcrt_knn's per-lane branch does)
        const int rc = emu_knn_cap(map, n, p, k, cap, out_count, out_index, out_d2);
        if (rc != 1 || cap >= kMaxVisitLimit) return rc == 1 ? -300 : rc;
    }
}
int emu_knn_cap(const mcrt_photon_map_desc* map, uint64_t n, const double* p, uint32_t k, uint32_t visit_cap, uint32_t* out_count, uint32_t* out_index,
                double* out_d2) {
    Emu E;
    PhotonMapView v;
    setupMap(E, 0, map, v);
    setupKnn(E, k, visit_cap ? visit_cap : kMaxVisit);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t visits = 0;
        uint32_t c = knnSearch(v, ld3(p + 3 * i), k, E.ks, visits);
        if (E.ks.overflowed) return 1;
        out_count[i] = c;
        std::vector<std::pair<double, uint32_t>> r;
        for (uint32_t q = 0; q < c; q++) r.push_back({E.ks.res(q).distance2, E.ks.res(q).index});
        std::sort(r.begin(), r.end());
        for (uint32_t q = 0; q < k; q++) {
            out_index[i * k + q] = q < c ? r[q].second : 0xFFFFFFFFu;
            out_d2[i * k + q] = q < c ? r[q].first : INFINITY;
        }
    }
    return 0;
}

// mcrt_output.hip's kernels as loops: statKernel / histKernel / levelKernel per pass, then developKernel — the per-pixel
// functions are the product's (mcrt_output.hpp).
int emu_tonemap(const double* rgb, uint32_t width, uint32_t height, uint32_t tonemapper, int plain, double exposure_ev, double gain_ev,
                uint8_t* bgr, double* factors) {
    const uint64_t n = (uint64_t)width * height;
    double factor[2] = {1.0, 1.0};
    if (!plain) {
        const double scale[2] = {std::pow(2.0, exposure_ev), std::pow(2.0, gain_ev)}, pct[2] = {0.5, 0.99};
        for (int pass = 0; pass < 2; pass++) {
            auto brightness = [&](uint64_t i) {
                const d3 p{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
                if (pass == 0) return brightnessOf(p);
                return brightnessOf(tonemapApply(tonemapper, false, d3{p.x * factor[0], p.y * factor[0], p.z * factor[0]}));
            };
            unsigned long long mx = 0;
            bool neg = false;
            for (uint64_t i = 0; i < n; i++) {
                const double v = brightness(i);
                if (v < 0.0) neg = true;
                if (v > 0.0) mx = std::max(mx, dBits(v));
            }
            double level = 0.0;
            const double m = bitsD(mx);
            if (!neg && m > 0.0) {
                std::vector<uint32_t> hist(kHistogramBins, 0);
                const double bin_size = m / (double)kHistogramBins;
                for (uint64_t i = 0; i < n; i++) hist[histogramBin(brightness(i), bin_size)]++;
                const unsigned long long num = (unsigned long long)((double)n * pct[pass]);
                unsigned long long count = 0;
                for (uint32_t b = 0; b < kHistogramBins; b++) {
                    count += hist[b];
                    if (count >= num) {
                        level = (double)(b + 1) * bin_size;
                        break;
                    }
                }
            }
            factor[pass] = (level > 0.0 ? pct[pass] / level : 1.0) * scale[pass];
        }
    }
    for (uint64_t i = 0; i < n; i++)
        developPixel(tonemapper, plain != 0, d3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, factor[0], factor[1], bgr + 3 * i);
    factors[0] = factor[0];
    factors[1] = factor[1];
    return 0;
}

// mcrt_plan.hpp: out = {pass_rows, store_bytes, chunk_shift, chunk}
void emu_plan(uint32_t width, uint32_t owned_rows, uint32_t spp, double store_gb, uint64_t want_units, uint64_t* out) {
    const PassPlan pp = planPasses(width, owned_rows, spp, store_gb);
    const ChunkPlan cp = planChunks(spp, want_units);
    out[0] = pp.pass_rows;
    out[1] = pp.store_bytes;
    out[2] = cp.shift;
    out[3] = cp.chunk;
}

uint64_t emu_plan_pool_slots(uint64_t pass_paths, uint64_t max_slots, uint64_t block) { return planPoolSlots(pass_paths, max_slots, block); }

// planChunksMega (mcrt_plan.hpp): out = {chunk_shift, chunk}
void emu_plan_mega(uint32_t spp, uint64_t lanes, uint64_t pass_pixels, uint64_t* out) {
    const ChunkPlan cp = planChunksMega(spp, lanes, pass_pixels);
    out[0] = cp.shift;
    out[1] = cp.chunk;
}

}  // extern "C"
