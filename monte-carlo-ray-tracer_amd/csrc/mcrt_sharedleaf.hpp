// The trace kernels' shared leaf step and the wave-synchronous tree walks built from the lane state machine's step functions
// (mcrt_lanesm.hpp, mcrt_qbvh.hpp). Device code; included by mcrt_kernels.hpp inside the kernels' namespace, and - unchanged - by the
// host emulation of a wavefront (tests/emu/wave_emu.hpp, tests/emu/wave_walk_emu.cpp), which is why it is a file of its own.
#pragma once

// ---- Shared leaf step (round 4): the wave's pending leaves tested by ALL of its lanes ----------------------------------------
// What the counters said about the leaf step (C3, round 3): it runs with ~23 of a wave's 64 lanes, every one of them testing TWO
// primitives in sequence (two 80-byte records, ~300 instructions with the hit updates) - 35 % of the kernel's wave cycles at a
// third of the lanes. Here the work items of a leaf step are (pending lane, primitive) PAIRS and they are dealt over all 64 lanes:
// every pending lane offers up to four primitives of its leaf, the offers are numbered by a prefix sum over the lanes and the first 64
// are the step's items: item lane k tests the j-th offered primitive of its owner with the owner's ray - the ray (start, direction: twelve
// 32-bit words) and the leaf range are PULLED from the owner with ds_bpermute, no LDS is written except a 64-byte rank -> lane map
// per wave. The owner then pulls the entry distances of its item lanes back, keeps the FIRST minimum (items are in ascending
// primitive order, so ties go to the lowest index: the tie rule of `closer`), pulls that item's u / v and updates its hit exactly as
// travPendStep does. Same tests (primTestRec: the reference's FP64 arithmetic), same minimum, same tie rule: the hit is the one
// every other form returns. A leaf of up to four primitives is one step instead of two, a step costs one primitive test instead of
// two, and it is worth issuing with far fewer pending lanes (the gate MCRT_WF_LEAF drops from 24 to 16), so lanes wait less at
// their leaves. Shadow queries: the winner is the closest accepted item, so "an occluder closer than t_near" is seen on the winner
// (an occluder that is not the winner has a closer one in front of it).
constexpr uint32_t kShareMapBytes = 64;  // per wave
__device__ __forceinline__ uint32_t wavePull(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
__device__ __forceinline__ double wavePullD(double v, uint32_t src_lane) {
    const unsigned long long b = dBits(v);
    const uint32_t lo = wavePull((uint32_t)b, src_lane), hi = wavePull((uint32_t)(b >> 32), src_lane);
    return bitsD(((unsigned long long)hi << 32) | lo);
}
// Items: every pending lane offers up to kShareCap primitives of its leaf; the offers are numbered by an exclusive prefix sum over the
// lanes (the counts are 0 .. 4: three bit planes, one ballot and one mbcnt each) and the first 64 are a step's. The kernel's gate looks
// at the TOTAL (a step is worth issuing when it fills the wave), so the offers are made before the gate.
constexpr uint32_t kShareCap = 4u;
struct ShareOffer {
    uint32_t want;                    // this lane's offer (0: no pending leaf)
    unsigned long long b0, b1, b2;    // bit planes of the offers over the wave
    uint32_t total;                   // sum of the offers
};
__device__ __forceinline__ ShareOffer shareOffer(bool pend, const PendLeaf& P) {
    ShareOffer s;
    s.want = pend ? (P.n < kShareCap ? P.n : kShareCap) : 0u;
    s.b0 = waveBallot((s.want & 1u) != 0u);
    s.b1 = waveBallot((s.want & 2u) != 0u);
    s.b2 = waveBallot((s.want & 4u) != 0u);
    s.total = (uint32_t)__popcll(s.b0) + 2u * (uint32_t)__popcll(s.b1) + 4u * (uint32_t)__popcll(s.b2);
    return s;
}
template <bool kCount>
__device__ __forceinline__ void travSharedLeafStep(const SmSceneView<false>& sv, Trav& T, PendLeaf& P, const ShareOffer& so,
                                                    MCRT_LDS_AS uint8_t* map, TraceCounters& cnt) {
    const uint32_t lane = laneId();
    const uint32_t want = so.want, total = so.total;
    auto below = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    const uint32_t pre = below(so.b0) + 2u * below(so.b1) + 4u * below(so.b2);
    const uint32_t take = pre >= 64u ? 0u : (want < 64u - pre ? want : 64u - pre);  // what of this lane's offer fits
#pragma unroll
    for (uint32_t jj = 0u; jj < kShareCap; jj++)
        if (jj < take) map[pre + jj] = (uint8_t)lane;  // item -> owner
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool item = lane < (total < 64u ? total : 64u);
    const uint32_t src = item ? (uint32_t)map[lane] : lane;
    __builtin_amdgcn_wave_barrier();  // (the next step's writes stay behind these reads)
    const uint32_t pa = wavePull(P.a, src), pre_src = wavePull(pre, src);
    d3 o, d;
    o.x = wavePullD(T.o.x, src); o.y = wavePullD(T.o.y, src); o.z = wavePullD(T.o.z, src);
    d.x = wavePullD(T.d.x, src); d.y = wavePullD(T.d.y, src); d.z = wavePullD(T.d.z, src);
    Hit h;
    h.t = 0.0; h.u = 0.0; h.v = 0.0; h.surface = kNoSurface; h.interpolate = false;
    bool ok = false;
    if (item) {
        const PrimRec rec = loadPrim(sv.prim + (size_t)(pa + (lane - pre_src)) * kPrimStride);
        Ray r;
        r.start = o;
        r.direction = d;
        r.inv_direction = d3{0.0, 0.0, 0.0};
        if (rec.v[9] == 3.0) r.inv_direction = rcp3(d);  // Quadric::intersect clips to its box first (the same rcp3 the owner's travBegin took)
        r.medium_ior = 1.0; r.refraction_scale = 1.0; r.refraction_level = 0; r.depth = 0; r.diffuse_depth = 0; r.dirac_delta = false; r.refraction = false;
        if (kCount) cnt.prim_tests++;
        ok = primTestRec<true>(rec, r, h);
    }
    const double key = ok ? h.t : INFINITY;
    // owner side: the first minimum over its item lanes (ascending primitive index: ties go to the lowest)
    double win_t = INFINITY;
    uint32_t win_j = 0u;
#pragma unroll
    for (uint32_t jj = 0u; jj < kShareCap; jj++) {
        const double tj = wavePullD(key, (pre + jj) & 63u);
        if (jj < take && tj < win_t) {
            win_t = tj;
            win_j = jj;
        }
    }
    const uint32_t wl = (pre + win_j) & 63u;
    const double win_u = wavePullD(h.u, wl), win_v = wavePullD(h.v, wl);
    const uint32_t win_i = wavePull(h.interpolate ? 1u : 0u, wl);
    if (take != 0u) {
        bool decided = false;
        const uint32_t idx = P.a + win_j;
        if (win_t < INFINITY && closer(win_t, idx, T.best)) {
            T.best.t = win_t;
            T.best.u = win_u;
            T.best.v = win_v;
            T.best.interpolate = win_i != 0u;
            T.best.surface = idx;
            if (T.shadow && idx != T.light && win_t < T.t_near) decided = true;  // occluded for sure
        }
        if (decided) {
            T.sp = 0;
            T.active = false;
            T.need_pop = false;
            P.n = 0u;
        } else {
            P.a += take;
            P.n -= take;
        }
    }
}

// Wave-synchronous walk over the quantised child blocks (mcrt_qbvh.hpp) with the step functions of the lane state
// machine: inner steps while any lane has one, a leaf step when enough lanes wait at a leaf or nothing else is left.
// Used by the photon-mapping eye pass for trees that stay in HBM.
template <bool kCount>
__device__ inline Hit traceWalkQ(const SmSceneView<false>& sv, const QView<true>& qv, const SmStack& stk, const Ray& ray, bool shadow,
                                 const ShadowQuery* sq, TraceCounters& cnt) {
    Trav T;
    travBeginQ<false, true, kCount>(sv, qv, T, ray.start, ray.direction, ray.inv_direction, shadow, sq, cnt);
    for (;;) {
        const bool inner = T.active && (T.node_m & kSmInner);
        if (inner && T.fast) travInnerStepQ<true, kCount>(qv, T, stk, cnt);
        if (inner && !T.fast) travInnerStep<false, kCount>(sv, T, stk, cnt);
        const bool leaf = T.active && !(T.node_m & kSmInner);
        const unsigned long long m_leaf = waveBallot(leaf), m_inner = waveBallot(T.active && (T.node_m & kSmInner));
        if (!(m_leaf | m_inner)) break;
        if (m_leaf && (__popcll(m_leaf) >= 32 || __popcll(m_inner) < 8)) {
            if (leaf) travLeafStep<false, kCount>(sv, T, stk, cnt);
        }
    }
    return T.best;
}

// The same walk with the trace kernel's round-4 machinery (deferred leaves tested by the whole wave, one pop site, the stack's top in
// registers): EVERY lane of the wave calls it - `valid` says whether the lane has a ray - because the shared leaf step deals its
// primitive tests over all 64 lanes. `map`: 64 bytes of this wave's LDS that nothing else uses during the walk.
template <bool kCount>
__device__ inline Hit traceWalkShared(const SmSceneView<false>& sv, const QView<true>& qv, const SmStack& stk, bool valid, const Ray& ray, bool shadow,
                                      const ShadowQuery* sq, TraceCounters& cnt, MCRT_LDS_AS uint8_t* map) {
    Trav T;
    PendLeaf P;
    T.active = false;
    T.need_pop = false;
    T.sp = 0;
    T.shadow = false;
    T.fast = true;
    T.light = kNoSurface;
    T.t_near = 0.0;
    T.node_a = T.node_m = 0u;
    hitInit(T.best, kDblMax);
    T.o = T.d = T.inv = d3{0.0, 0.0, 0.0};
    if (valid) travBeginQ<false, true, kCount>(sv, qv, T, ray.start, ray.direction, ray.inv_direction, shadow, sq, cnt);
    for (;;) {
        if (T.active && !(T.node_m & kSmInner) && P.n == 0u) {
            P.a = T.node_a;
            P.n = T.node_m;
            T.active = false;
            T.need_pop = true;
        }
        if (waveBallot(T.need_pop)) {
            if (T.need_pop) {
                travPopCached(T, stk);
                T.need_pop = false;
            }
        }
        const bool inner = T.active && (T.node_m & kSmInner);
        if (inner && T.fast) travInnerStepQ<true, kCount, true>(qv, T, stk, cnt);
        if (inner && !T.fast) travInnerStep<false, kCount, true>(sv, T, stk, cnt);
        const bool pend = P.n != 0u;
        const unsigned long long m_pend = waveBallot(pend);
        if (!(m_pend | waveBallot(T.need_pop || T.active))) break;
        if (m_pend) {
            const unsigned long long m_inner = waveBallot(T.need_pop || (T.active && (T.node_m & kSmInner)));
            const ShareOffer so = shareOffer(pend, P);
            if (so.total >= 48u || __popcll(m_inner) < 8) travSharedLeafStep<kCount>(sv, T, P, so, map, cnt);
        }
    }
    return T.best;
}

