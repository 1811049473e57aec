// Per-lane STATE MACHINE form of the path tracer, for scenes whose BVH has to be walked (everything
// that is not the tiny-scene flat loop).
//
// Why: in a BVH walk every lane follows its own node sequence and rays need very different numbers
// of steps (spaceship.json: 34 node visits per ray on average, >150 for some). When a whole wave runs
// "traverse until every lane is done, then shade", most lanes idle most of the time (measured VALU lane
// utilisation 14 %, profiles/). Here each lane carries an explicit state
//     REGEN -> TRAV(bounce ray) -> SHADE -> TRAV(shadow ray) -> [NEE finish] -> TRAV(next bounce ray) -> SHADE ...
// and the wave executes ONE short block per loop iteration for the lanes that are in the matching
// state: an inner-node step, a leaf step, or (only when enough lanes have queued up for it, or nothing
// else can run) the expensive shade / regenerate blocks. Lanes that finish a traversal early start
// their next one immediately instead of waiting for the slowest ray of the wave.
//
// The arithmetic of a path is exactly that of mcrt_integrator.hpp (same functions, same order); only
// the interleaving between lanes changes. The next-event estimate is split around the shadow
// traversal: everything that does not depend on the shadow hit (BSDF value/pdf towards the light,
// area*cos of the light, throughput at that bounce) is computed in the shade block and parked in
// NeePending, so the Interaction does not have to stay live across a traversal.
//
// Traversal data: 64-byte node records {bounds[6], a, b} with the children of a node contiguous, so a
// child's meta arrives with its box and a node visit is ONE dependent memory round trip; stack
// entries are 8 bytes {entry-distance key, a} with the node's flag/count packed into the low 9 bits
// of the (rounded-down, therefore still conservative) float key.
#pragma once

#include "mcrt_integrator.hpp"

namespace mcrt {

template <bool kAll>
struct SmSceneView {
    uint32_t num_nodes;
    cptr<Node64, kAll> nodes;
    cptr<double, kAll> prim;
    uint32_t lds_nodes;  // nodes [0, lds_nodes) are also in LDS (used when !kAll)
    MCRT_LDS_AS const Node64* lds_node_ptr;
};

struct SmStackEntry {
    uint32_t key;  // float bits of floor_f32(entry t) with the low 9 bits replaced by the node's m
    uint32_t a;
};

// sp * stride of a stack slot: both are far below 2^24, and the 24-bit multiply issues at full rate where v_mul_lo_u32 takes four
// slots (a block visit pushes up to four entries and the loop pops one: five multiplies per iteration of the trace kernel)
MCRT_HD uint32_t stackSlot(int sp, uint32_t stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24((uint32_t)sp, stride);
#else
    return (uint32_t)sp * stride;
#endif
}

struct SmStack {
    MCRT_LDS_AS SmStackEntry* lds;
    uint32_t lds_stride;
    SmStackEntry* spill;
    uint32_t spill_stride;
    int lds_depth = kLdsStackDepth;  // entries per lane kept in LDS (the trace kernel trades some of them for more tree blocks)
    int max_depth = kMaxStackDepth;  // entries per lane in all: the scene's stack bound, at least kMaxStackDepth (HostLayout::stack_bound)
    // (the empty asm statements keep the two address spaces in separate branches: merged into one generic-pointer access the
    // compiler emits flat_load / flat_store, which wait on both the LDS and the vector-memory counters)
    MCRT_HD void put(int sp, SmStackEntry e) const {
        if (sp < lds_depth) {
            lds[stackSlot(sp, lds_stride)] = e;
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");
#endif
            spill[(uint32_t)(sp - lds_depth) * (size_t)spill_stride] = e;
        }
    }
    MCRT_HD SmStackEntry get(int sp) const {
        SmStackEntry e;
        if (sp < lds_depth) {
            e = lds[stackSlot(sp, lds_stride)];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(e.key), "+v"(e.a));  // the value exists HERE: the load cannot sink below the join
#endif
        } else {
            e = spill[(uint32_t)(sp - lds_depth) * (size_t)spill_stride];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(e.key), "+v"(e.a));
#endif
        }
        return e;
    }
};

// One closest-hit query in progress.
struct Trav {
    d3 o, d, inv;      // the ray being traversed
    Hit best;
    uint32_t node_a, node_m;  // meta of the node to visit next
    int sp;
    bool active;       // a node is waiting to be visited
    bool need_pop = false;  // (lazy-pop walks, wfTraceKernel forms 2 / 3) the next node is still on the stack: popped at the loop's one pop site
    // (same walks) the stack's TOP entry, also held in registers: every push leaves the entry it wrote last here (write-through: it
    // is in the stack's memory too), a pop takes it without waiting for an LDS read and asks for the entry below at once - that read
    // lands while the wave runs its next step. top_key == kTopNone: not cached, read the stack.
    uint32_t top_key = 0xFFFFFFFFu, top_a = 0u;
    bool fast;         // v_min/v_max box test allowed (no NaN slab products possible)
    bool shadow;
    uint32_t light;    // shadow query: surface aimed at
    double t_near;     // shadow query: a closer hit of another surface decides it
};

// What travInnerStepQLean (mcrt_qbvh.hpp) keeps per ray instead of recomputing it at every visit: the ray rounded to FP32 as
// travInnerStepQ rounds it, and the smallest float >= best.t. Its own struct, held by the trace kernel only: as members of Trav the
// seven words cost renderKernelPM - whose Trav lives partly in scratch - 8 % of a C5 frame (round 5, profiles/r05_ab_c5_bisect.log).
struct LeanRay {
    float of[3], invf[3], best_up;
};
MCRT_HD void leanRayBegin(LeanRay& R, const Trav& T) {
    R.of[0] = (float)T.o.x; R.of[1] = (float)T.o.y; R.of[2] = (float)T.o.z;
    R.invf[0] = (float)T.inv.x; R.invf[1] = (float)T.inv.y; R.invf[2] = (float)T.inv.z;
    R.best_up = floatAbove(T.best.t);
}

template <bool kAll>
MCRT_HD Box smLoadBox(const SmSceneView<kAll>& sv, uint32_t i, uint32_t& a, uint32_t& m) {
    Box b;
    if (!kAll && i < sv.lds_nodes) {
        MCRT_LDS_AS const Node64* n = sv.lds_node_ptr + i;
        for (int k = 0; k < 6; k++) b.v[k] = n->b[k];
        a = n->a;
        m = n->m;
    } else {
        cptr<Node64, kAll> n = sv.nodes + i;
        for (int k = 0; k < 6; k++) b.v[k] = n->b[k];
        a = n->a;
        m = n->m;
    }
    return b;
}

MCRT_HD Ray travRay(const Trav& T) {  // only start/direction/inv_direction are read by the tests
    Ray r;
    r.start = T.o;
    r.direction = T.d;
    r.inv_direction = T.inv;
    r.medium_ior = 1.0;
    r.refraction_scale = 1.0;
    r.refraction_level = 0;
    r.depth = 0;
    r.diffuse_depth = 0;
    r.dirac_delta = false;
    r.refraction = false;
    return r;
}

constexpr uint32_t kTopNone = 0xFFFFFFFFu;  // (never a stack key: the block visit's "miss" sentinel is not pushed)
// The pop of the lazy-pop walks: top entry from registers, the next one requested before the popped one is looked at. Written as a
// straight-line first attempt (the popped entry is visited nine times in ten) so that the request for the entry below goes into the
// lane's own top registers with nothing waiting for it here; only a culled entry falls into the loop, which does wait.
MCRT_HD void travTopRequest(Trav& T, const SmStack& stk) {  // T.sp > 0: ask for stack[sp - 1]
    const int below = T.sp - 1;
    if (below < stk.lds_depth) {
        const SmStackEntry n = stk.lds[stackSlot(below, stk.lds_stride)];
        T.top_key = n.key;
        T.top_a = n.a;
    } else {
        const SmStackEntry n = stk.get(below);
        T.top_key = n.key;
        T.top_a = n.a;
    }
}
MCRT_HD void travPopCached(Trav& T, const SmStack& stk) {
    T.active = false;
    if (T.sp <= 0) return;
    SmStackEntry e;
    if (T.top_key != kTopNone) {
        e.key = T.top_key;
        e.a = T.top_a;
    } else {
        e = stk.get(T.sp - 1);
    }
    --T.sp;
    T.top_key = kTopNone;
    if (T.sp > 0) travTopRequest(T, stk);
    if ((double)bitsFloat(e.key & ~0x1FFu) <= T.best.t) {
        T.node_a = e.a;
        T.node_m = e.key & 0x1FFu;
        T.active = true;
        return;
    }
    // The popped entry was culled (a hit since it was pushed): work down the stack, one entry per round trip. (Four entries per trip -
    // a lane's entries at consecutive depths are independent LDS reads - was measured SLOWER, C3 393.9 -> 400.3 ms: culled runs are
    // short, and the extra reads and selects cost every pop that reaches this loop.)
    while (T.sp > 0) {
        e.key = T.top_key;
        e.a = T.top_a;
        --T.sp;
        T.top_key = kTopNone;
        if (T.sp > 0) travTopRequest(T, stk);
        if ((double)bitsFloat(e.key & ~0x1FFu) <= T.best.t) {
            T.node_a = e.a;
            T.node_m = e.key & 0x1FFu;
            T.active = true;
            break;
        }
    }
}

MCRT_HD void travPop(Trav& T, const SmStack& stk) {
    T.active = false;
    while (T.sp > 0) {
        SmStackEntry e = stk.get(--T.sp);
        if ((double)bitsFloat(e.key & ~0x1FFu) <= T.best.t) {
            T.node_a = e.a;
            T.node_m = e.key & 0x1FFu;
            T.active = true;
            break;
        }
    }
}

// One primitive record in registers.
struct PrimRec {
    double v[kPrimStride];
};
template <class P>
MCRT_HD PrimRec loadPrim(P p) {
    PrimRec r;
    for (int k = 0; k < kPrimStride; k++) r.v[k] = p[k];
    return r;
}
template <bool kQuadrics>
MCRT_HD bool primTestRec(const PrimRec& rec, const Ray& ray, Hit& h) {
    if (rec.v[9] == 1.0 || (kQuadrics && rec.v[9] == 3.0)) return primIntersect<kQuadrics>(rec.v, ray, h);  // sphere / quadric (rare in walked BVHs)
    double t, u, v;
    const bool ok = triangleTestFlat(rec.v, ray.start, ray.direction, t, u, v);
    const bool interp = rec.v[9] >= 2.0;
    h.t = t;
    h.u = interp ? u : 0.0;
    h.v = interp ? v : 0.0;
    h.interpolate = interp;
    return ok;
}

// Scene::intersect / BVH::intersect start (bvh.cpp:84-88): root box test.
template <bool kAll, bool kCount>
MCRT_HD void travBegin(const SmSceneView<kAll>& sv, Trav& T, d3 start, d3 direction, d3 inv_direction, bool shadow,
                       const ShadowQuery* sq, TraceCounters& cnt) {
    T.o = start;
    T.d = direction;
    T.inv = inv_direction;
    T.best.t = shadow ? sq->t_far : kDblMax;
    T.best.u = 0.0;
    T.best.v = 0.0;
    T.best.surface = kNoSurface;
    T.best.interpolate = false;
    T.shadow = shadow;
    T.light = shadow ? sq->light : kNoSurface;
    T.t_near = shadow ? sq->t_near : 0.0;
    T.sp = 0;
    T.need_pop = false;
    T.top_key = kTopNone;
    // (1e25: the FP32 slab test of the quantised blocks, mcrt_qbvh.hpp, multiplies scene-sized lengths by these)
    T.fast = fabs(inv_direction.x) <= 1e25 && fabs(inv_direction.y) <= 1e25 && fabs(inv_direction.z) <= 1e25;
    cnt.rays++;
    const Ray r = travRay(T);
    if (shadow) {
        // Shadow query, exactly (integrator.cpp:68-73: the closest hit must BE the light): the light's own intersection first - the
        // reference's FP64 test on the light's record - then "is any other surface closer than THAT" with the tie rule of every other
        // query. (Until round 4 the bound was the distance d to the sampled point, d (1 +- 1e-9): off when the ray grazes the light - a
        // shading point in the plane of a light triangle, lego_bulldozer's lamp quads - where the computed t deviates from d by more than
        // that, and a frame lost the reference's last bits.) A light the exact test does not hit cannot be the closest hit: no walk at all.
        Hit h;
        if (kCount) cnt.prim_tests++;
        const PrimRec rec = loadPrim(sv.prim + (size_t)sq->light * kPrimStride);
        if (!primTestRec<QuadricsIn<kAll>::value>(rec, r, h)) {
            T.active = false;
            T.node_a = T.node_m = 0u;
            return;
        }
        // the light IS the hit so far (its leaf's box may start an ulp behind this t and be culled: the walk need not find it again);
        // another surface replaces it when it is closer - or exactly as far with a lower index, the tie rule of every query
        T.best = h;
        T.best.surface = sq->light;
        T.t_near = h.t;
    }
    if (kCount) cnt.node_tests++;
    uint32_t a, m;
    Box rb = smLoadBox(sv, 0u, a, m);
    double t;
    T.active = T.fast ? boxIntersect<true>(rb, r, t) : boxIntersect<false>(rb, r, t);
    T.node_a = a;
    T.node_m = m;
}

// Visit one INNER node: test its children (bvh.cpp:108-119), continue with the nearest hit child, push the rest.
template <bool kAll, bool kCount, bool kLazyPop = false>
MCRT_HD void travInnerStep(const SmSceneView<kAll>& sv, Trav& T, const SmStack& stk, TraceCounters& cnt) {
    const uint32_t first = T.node_a, count = T.node_m & 0xFFu;
    const Ray r = travRay(T);
    double near_t = 0.0;
    uint32_t near_a = 0, near_m = 0;
    bool have_near = false;
    for (uint32_t c = first; c < first + count; c++) {
        uint32_t a, m;
        Box cb = smLoadBox(sv, c, a, m);
        if (kCount) cnt.node_tests++;
        double t;
        const bool hit = T.fast ? boxIntersect<true>(cb, r, t) : boxIntersect<false>(cb, r, t);
        if (hit && t <= T.best.t) {
            uint32_t push_a = a, push_m = m;
            double push_t = t;
            bool push = true;
            if (!have_near || t < near_t) {
                push = have_near;
                push_a = near_a;
                push_m = near_m;
                push_t = near_t;
                near_a = a;
                near_m = m;
                near_t = t;
                have_near = true;
            }
            if (push) {
                if (T.sp < stk.max_depth) {
                    SmStackEntry e;
                    e.key = (floatBits(floatBelow(push_t)) & ~0x1FFu) | push_m;
                    e.a = push_a;
                    stk.put(T.sp++, e);
                    T.top_key = e.key;  // (the stack's top, cached: travPopCached)
                    T.top_a = e.a;
                } else {
                    cnt.overflow = 1;
                }
            }
        }
    }
    if (have_near) {
        T.node_a = near_a;
        T.node_m = near_m;
    } else if (kLazyPop) {
        T.active = false;
        T.need_pop = true;
    } else {
        travPop(T, stk);
    }
}

// Tests the next primitive(s) of the range [i, i + count) against the ray of T with the reference's FP64 tests; returns how many
// were consumed (1 or 2). (Round 3 put an FP32 cull of an aligned primitive pair in front - the flat loop's records per leaf pair;
// compiled into the gfx950 kernels the branch cost 4-5 % of a frame, it never shipped enabled and was removed in round 6.)
template <bool kAll, bool kCount>
MCRT_HD uint32_t leafTestNext(const SmSceneView<kAll>& sv, Trav& T, uint32_t i, uint32_t count, TraceCounters& cnt, bool& decided) {
    const Ray r = travRay(T);
    const bool two = count > 1u;
    const uint32_t j = two ? i + 1 : i;
    const PrimRec r0 = loadPrim(sv.prim + (size_t)i * kPrimStride);
    const PrimRec r1 = loadPrim(sv.prim + (size_t)j * kPrimStride);
    Hit h0, h1;
    if (kCount) cnt.prim_tests += two ? 2u : 1u;
    const bool ok0 = primTestRec<QuadricsIn<kAll>::value>(r0, r, h0);
    const bool ok1 = primTestRec<QuadricsIn<kAll>::value>(r1, r, h1) && two;
    if (ok0 && closer(h0.t, i, T.best)) {
        T.best = h0;
        T.best.surface = i;
        if (T.shadow && i != T.light && h0.t < T.t_near) decided = true;  // occluded for sure
    }
    if (ok1 && closer(h1.t, j, T.best)) {
        T.best = h1;
        T.best.surface = j;
        if (T.shadow && j != T.light && h1.t < T.t_near) decided = true;
    }
    return two ? 2u : 1u;
}

// One step at a LEAF: test its next (pair of) primitives (bvh.cpp:92-107); the lane stays at the leaf while primitives are
// left, then pops. A step is ONE pair, not the whole leaf: the lanes of a wave sit at leaves of different sizes, and a loop over
// the whole leaf keeps the lanes of the small ones idle until the largest is done.
template <bool kAll, bool kCount>
MCRT_HD void travLeafStep(const SmSceneView<kAll>& sv, Trav& T, const SmStack& stk, TraceCounters& cnt) {
    const uint32_t i = T.node_a, count = T.node_m;
    bool decided = false;
    const uint32_t used = leafTestNext<kAll, kCount>(sv, T, i, count, cnt, decided);
    if (decided) {
        T.sp = 0;
        T.active = false;
    } else if (count > used) {
        T.node_a = i + used;
        T.node_m = count - used;
    } else {
        travPop(T, stk);
    }
}

// ---- Deferred leaves (round 3) ---------------------------------------------------------------------------------------------
// The lanes of a wave reach leaves at different moments, and a leaf step (FP64 primitive tests) is only worth issuing when many
// lanes take part - so a lane at a leaf WAITS (idle through the other lanes' inner steps) until enough lanes have arrived. In the
// wavefront pipeline's trace kernel (MCRT_WF_DEFER, default 1 since round 3) a lane that reaches a leaf PARKS it (one pending leaf per lane, two registers) and keeps walking: it
// pops its next node and takes part in the following inner steps; the pending leaves of the wave are tested together once enough
// lanes have one, or when few lanes are left with inner nodes to visit. The closest hit is a minimum over exact FP64 primitive
// tests with the lowest-index tie rule, so it does not depend on WHEN a leaf is tested; what changes is pruning: nodes visited
// while a leaf is pending are not yet cut off by that leaf's hit. Measured on C3: + 3.4 % box tests, + 4 % primitive
// tests; compiled next to the waiting form in one kernel it was neutral, as its own lean kernel instance it takes 1.3 % off a C3 and
// a C4 frame (the waiting form remains as MCRT_WF_DEFER=0 and in the megakernels). The pending-leaf slot is also what the
// eight-wide walk (mcrt_wbvh.hpp) keeps its current leaf in.
struct PendLeaf {
    uint32_t a = 0, n = 0;  // first primitive, primitives left (0: none pending)
};

// The lane stands at a leaf and its pending slot is free: park the leaf, move on to the next node of the stack.
MCRT_HD void travParkLeaf(Trav& T, PendLeaf& P, const SmStack& stk) {
    if (T.active && !(T.node_m & kSmInner) && P.n == 0u) {
        P.a = T.node_a;
        P.n = T.node_m;
        travPop(T, stk);
    }
}

// One step on the pending leaf: its next (pair of) primitives, as travLeafStep.
template <bool kAll, bool kCount>
MCRT_HD void travPendStep(const SmSceneView<kAll>& sv, Trav& T, PendLeaf& P, TraceCounters& cnt) {
    const uint32_t i = P.a, count = P.n;
    if (count == 0u) return;
    bool decided = false;
    const uint32_t used = leafTestNext<kAll, kCount>(sv, T, i, count, cnt, decided);
    if (decided) {
        T.sp = 0;
        T.active = false;
        T.need_pop = false;
        P.n = 0u;
    } else {
        P.a = i + used;
        P.n = count - used;
    }
}

// What the next-event estimate of a bounce needs once its shadow ray has been traced.
struct NeePending {
    bool pending;  // a shadow ray is being traced for this bounce
    uint32_t light;
    d3 bsdf_absIdotN;
    double bsdf_pdf, area_cos;
    d3 throughput;  // throughput the estimate is weighted with (before this bounce's BSDF update)
};

// Second half of Integrator::sampleDirect (integrator.cpp:68-86) from the parked data.
// light_material: sh.surf_material[nee.light], for callers that asked for it ahead of time (mcrt_wavefront.hpp)
template <bool L>
MCRT_HD void smNeeFinish(PathState& st, const ShadeViewT<L>& sh, const NeePending& nee, const Hit& shadow_hit, uint32_t light_material) {
    if (shadow_hit.surface == kNoSurface || shadow_hit.surface != nee.light) return;
    double light_pdf = sq(shadow_hit.t) / nee.area_cos;
    double mis_weight = powerHeuristic(light_pdf, nee.bsdf_pdf);
    const auto& lm = sh.materials[light_material];
    d3 direct = mis_weight * nee.bsdf_absIdotN * ld3(lm.emittance) / (light_pdf * st.ls.select_probability);
    st.radiance = st.radiance + direct * nee.throughput;
}
template <bool L>
MCRT_HD void smNeeFinish(PathState& st, const ShadeViewT<L>& sh, const NeePending& nee, const Hit& shadow_hit) {
    if (shadow_hit.surface == kNoSurface || shadow_hit.surface != nee.light) return;
    smNeeFinish(st, sh, nee, shadow_hit, sh.surf_material[nee.light]);
}

// The tail of a bounce once the Interaction exists: next-event estimate set-up (cut at its shadow-ray trace), BSDF
// sampling, throughput update, russian roulette, refraction history. Shared by the path tracer's shade block and the
// photon mapper's (photon-mapper.cpp:308-311,319-325,334-339), which skips the next-event estimate on dirac hits.
template <bool L>
MCRT_HD bool smContinue(PathState& st, RefractionHistory& rh, const ShadeViewT<L>& sh, const InteractionT<L>& ia, bool do_nee, NeePending& nee,
                        Ray& shadow_ray, ShadowQuery& shadow_q, SobolTab tab) {
    nee.pending = false;
    DirectQuery dq;  // path-tracer.cpp:35 / photon-mapper.cpp:319-323, integrator.cpp:31-66
    if (do_nee && sampleDirectSetup(sh, ia, st.ls, dq, st.smp, tab)) {
        // integrator.cpp:75-81 evaluated ahead of the trace: it does not depend on the shadow hit
        d3 bsdf_absIdotN;
        double bsdf_pdf;
        if (interactionBSDF(ia, bsdf_absIdotN, dq.shadow_ray.direction, bsdf_pdf)) {
            nee.pending = true;
            nee.light = dq.sq.light;
            shadow_ray = dq.shadow_ray;
            shadow_q = dq.sq;
            nee.bsdf_absIdotN = bsdf_absIdotN;
            nee.bsdf_pdf = bsdf_pdf;
            nee.area_cos = (L ? sh.surf_area[st.ls.light] : dq.light_area) * dq.cos_light_theta;
            nee.throughput = st.throughput;
        }
    }

    d3 bsdf_absIdotN;
    if (!interactionSampleBSDF(ia, bsdf_absIdotN, st.ls.bsdf_pdf, st.ray, false, st.smp, tab)) return false;  // :37-40
    st.throughput = st.throughput * (bsdf_absIdotN / st.ls.bsdf_pdf);                                           // :42
    if (absorb(st.ray, st.throughput, st.smp, tab)) return false;                                               // :44-47
    rh.update(st.ray);                                                                                          // :49
    return true;
}

// The shade block: everything PathTracer::sampleRay does between two Scene::intersect calls of the
// path (path-tracer.cpp:27-49), with sampleDirect cut at its shadow-ray trace. Returns true when the
// path continues with st.ray; nee.pending says whether a shadow ray (returned in shadow_ray / shadow_q)
// must be traced first.
template <bool L>
MCRT_HD bool smShade(PathState& st, RefractionHistory& rh, const ShadeViewT<L>& sh, const Hit& isect, NeePending& nee,
                     Ray& shadow_ray, ShadowQuery& shadow_q, SobolTab tab) {
    nee.pending = false;
    if (isect.surface == kNoSurface) {  // :27-30
        st.radiance = st.radiance + skyColor(st.ray) * st.throughput;
        return false;
    }
    InteractionT<L> ia;
    interactionInit(ia, sh, isect, st.ray, rh.externalIOR(st.ray), st.smp, tab);  // :32
    st.radiance = st.radiance + sampleEmissive(sh, ia, st.ls) * st.throughput;   // :34

    return smContinue(st, rh, sh, ia, true, nee, shadow_ray, shadow_q, tab);
}

}  // namespace mcrt
