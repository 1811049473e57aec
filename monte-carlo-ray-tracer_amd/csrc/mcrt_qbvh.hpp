// Quantised child blocks for the trace kernel of the wavefront pipeline.
//
// Why: with the tree in HBM the trace kernel is bound by the number of 16-byte lane requests the CU's
// L1 (TCP) can look up, not by latency or arithmetic (measured on metal_bunnies: 266 requests per ray,
// TCP active 90 % of the time, VALU 33 %, no gain from more waves). A node visit with 64-byte FP64 child
// records costs 4 requests per child. Here the children of an inner node share ONE 64-byte block: a
// float origin, a power-of-two cell size per axis and 8-bit cell coordinates of every child box, plus the
// children's links — 4 requests per node visit instead of 16.
//
// Exactness: a decoded box is an FP64 box that CONTAINS the reference's FP64 box of that child (the
// host rounds lower bounds down and upper bounds up and verifies the decode with the device's own
// expression), and the slab test run on it is the reference's (BoundingBox::intersect, bounding-box.cpp:9-17).
// Every operation of that test is monotone in the bounds, so a decoded box is hit whenever the exact box
// is, with an entry distance that is not larger: the walk visits a superset of the nodes the exact walk
// visits and therefore tests a superset of the primitives. Primitive tests are unchanged (FP64,
// Triangle::intersect / Sphere::intersect) and the result is the minimum over the tested primitives, so
// it equals the exact walk's result. Rays with a zero direction component (0 * inf = NaN slabs, where
// "monotone" does not hold) never use the blocks; they walk the exact 64-byte records (Trav::fast == false).
#pragma once

#include "mcrt_lanesm.hpp"

namespace mcrt {

// 64 bytes, up to 4 children of one inner node:
//   w[0..2]   float origin x,y,z            (<= every child's lower bound)
//   w[3]      ex | ey << 8 | ez << 16 | n << 24     cell size per axis = 2^(e - 128); n = children in this
//             block (1..4) | 0x80 when the node's next block follows (node with more than 4 children)
//   w[4..9]   24 bytes, one word per (axis, side): w[4 + 2 axis + side] = the four children's lower (side 0) / upper (side 1)
//             cell coordinates on that axis, child c in byte c — so that the word of the planes a ray ENTERS through is
//             picked with one select per axis
//   w[10..13] child c: inner -> index of its first block, leaf -> first primitive
//   w[14..15] child c (16 bits each): inner -> 0x100 | number of children, leaf -> number of primitives
struct alignas(64) QBlock {
    uint32_t w[16];
};
constexpr int kQExpBias = 128;

MCRT_HD double qCell(uint32_t e) { return bitsD((unsigned long long)((int)e - kQExpBias + 1023) << 52); }
// The one decode expression, used by the host builder's verification and by the kernels. q * cell is exact (8 bits
// times a power of two), so the fused form rounds once, exactly like multiply-then-add would: one instruction.
MCRT_HD double qDecode(float origin, uint32_t q, double cell) { return fma((double)q, cell, (double)origin); }
MCRT_HD uint32_t qByte(const QBlock& b, int child, int axis, int side) { return (b.w[4 + 2 * axis + side] >> (8 * child)) & 0xFFu; }

template <bool kLds>
struct QView {
    const QBlock* blocks;                 // every block, HBM
    uint32_t lds_blocks;                  // blocks [0, lds_blocks) are also in LDS (top of the tree)
    MCRT_LDS_AS const QBlock* lds_ptr;
    uint32_t root_a, root_m;              // the root as a child: inner -> block 0, leaf -> its primitives
};

template <bool kLds>
MCRT_HD QBlock qFetch(const QView<kLds>& qv, uint32_t i) {
    QBlock b;
    if (kLds && i < qv.lds_blocks) {
        MCRT_LDS_AS const QBlock* p = qv.lds_ptr + i;
        for (int k = 0; k < 16; k++) b.w[k] = p->w[k];
    } else {
        const QBlock* p = qv.blocks + i;
        for (int k = 0; k < 16; k++) b.w[k] = p->w[k];
    }
    return b;
}

// Scene::intersect / BVH::intersect start (bvh.cpp:84-88): the root box is tested exactly (one record).
template <bool kAll, bool kLds, bool kCount>
MCRT_HD void travBeginQ(const SmSceneView<kAll>& sv, const QView<kLds>& qv, Trav& T, d3 start, d3 direction, d3 inv_direction,
                        bool shadow, const ShadowQuery* sq, TraceCounters& cnt) {
    travBegin<kAll, kCount>(sv, T, start, direction, inv_direction, shadow, sq, cnt);
    if (T.fast) {  // from here on node_a / node_m are block links
        T.node_a = qv.root_a;
        T.node_m = qv.root_m;
    }
}

// Visit one INNER node through its block(s): decode and test the children (bvh.cpp:108-119), continue with the nearest
// hit child, push the rest — farthest first, so that the stack hands them back nearest first (the closer a subtree is
// visited, the sooner T.best.t cuts the others off). The four children of a block are tested and ordered without
// branches (a 5-exchange sorting network on {entry distance, link}); only the pushes are conditional.
// (FP64 form on the decoded boxes — what the superset argument above is stated for; kept as the yardstick of the FP32
// form below in the host-emulation tests)
template <bool kLds, bool kCount>
MCRT_HD void travInnerStepQ64(const QView<kLds>& qv, Trav& T, const SmStack& stk, TraceCounters& cnt) {
    const Ray r = travRay(T);
    constexpr double kMiss = INFINITY;  // entry distance of a child that is absent, missed or culled
    double near_t = kMiss;
    uint32_t near_a = 0, near_m = 0;
    auto push = [&](double t, uint32_t a, uint32_t m) {
        if (T.sp < stk.max_depth) {
            SmStackEntry e;
            e.key = (floatBits(floatBelow(t)) & ~0x1FFu) | m;
            e.a = a;
            stk.put(T.sp++, e);
            T.top_key = e.key;
            T.top_a = e.a;
        } else {
            cnt.overflow = 1;
        }
    };
    uint32_t bi = T.node_a;
    bool more = true;
    while (more) {
        const QBlock b = qFetch(qv, bi++);
        const uint32_t n = (b.w[3] >> 24) & 0x7Fu;
        more = (b.w[3] >> 31) != 0u;
        const float ox = bitsFloat(b.w[0]), oy = bitsFloat(b.w[1]), oz = bitsFloat(b.w[2]);
        const double cx = qCell(b.w[3] & 0xFFu), cy = qCell((b.w[3] >> 8) & 0xFFu), cz = qCell((b.w[3] >> 16) & 0xFFu);
        double t[4];
        uint32_t a[4], m[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 4; c++) {
            Box cb;
            cb.v[0] = qDecode(ox, qByte(b, c, 0, 0), cx);
            cb.v[1] = qDecode(oy, qByte(b, c, 1, 0), cy);
            cb.v[2] = qDecode(oz, qByte(b, c, 2, 0), cz);
            cb.v[3] = qDecode(ox, qByte(b, c, 0, 1), cx);
            cb.v[4] = qDecode(oy, qByte(b, c, 1, 1), cy);
            cb.v[5] = qDecode(oz, qByte(b, c, 2, 1), cz);
            a[c] = b.w[10 + c];
            m[c] = (b.w[14 + c / 2] >> (16 * (c % 2))) & 0xFFFFu;
            double tt;
            const bool hit = boxIntersect<true>(cb, r, tt);
            const bool keep = (uint32_t)c < n && hit && tt <= T.best.t;
            if (kCount) cnt.node_tests += (uint32_t)c < n ? 1u : 0u;
            t[c] = keep ? tt : kMiss;
        }
        auto exchange = [&](int i, int j) {  // afterwards t[i] <= t[j]
            const bool sw = t[j] < t[i];
            const double ti = sw ? t[j] : t[i], tj = sw ? t[i] : t[j];
            const uint32_t ai = sw ? a[j] : a[i], aj = sw ? a[i] : a[j];
            const uint32_t mi = sw ? m[j] : m[i], mj = sw ? m[i] : m[j];
            t[i] = ti; t[j] = tj; a[i] = ai; a[j] = aj; m[i] = mi; m[j] = mj;
        };
        exchange(0, 1);
        exchange(2, 3);
        exchange(0, 2);
        exchange(1, 3);
        exchange(1, 2);
        if (t[3] < kMiss) push(t[3], a[3], m[3]);
        if (t[2] < kMiss) push(t[2], a[2], m[2]);
        if (t[1] < kMiss) push(t[1], a[1], m[1]);
        // the block's nearest against the nearest of the node's earlier blocks (nodes with more than 4 children)
        const bool better = t[0] < near_t;
        const double lose_t = better ? near_t : t[0];
        const uint32_t lose_a = better ? near_a : a[0], lose_m = better ? near_m : m[0];
        if (better) {
            near_t = t[0];
            near_a = a[0];
            near_m = m[0];
        }
        if (lose_t < kMiss) push(lose_t, lose_a, lose_m);
    }
    if (near_t < kMiss) {
        T.node_a = near_a;
        T.node_m = near_m;
    } else {
        travPop(T, stk);
    }
}

// The same visit with the slab tests in FP32 — the trace kernels' form. Per block and axis, with A = cell / d and
// C = (origin - o) / d, the plane of cell coordinate q is crossed at t = q A + C; the entry side uses C - m and the exit
// side C + m, where the margin m covers every rounding between the FP64 form above and the FP32 one (ray rounded to float,
// difference, products, multiply-add; the bound is derived where m is computed). So every FP32 entry distance is <= the
// FP64 one, every exit distance >=: a child the FP64 form keeps is kept here, with a key that is not larger — the walk
// visits a superset of the nodes again, and the result (a minimum over exact FP64 primitive tests) is unchanged. A
// child's {entry distance, link meta} travel as one word — the stack's key format (float bits, low 9 bits = m) — so the
// children are ordered with integer min / max and pushed as they are. (Round 2, second pass: A and C themselves in FP32 —
// six conversions of the ray per visit instead of nine FP64 operations and six conversions per block.)
constexpr uint32_t kQMissKey = 0xFFFFFFFFu;
template <bool kLds, bool kCount, bool kLazyPop = false>
MCRT_HD void travInnerStepQ(const QView<kLds>& qv, Trav& T, const SmStack& stk, TraceCounters& cnt) {
    const float best_up = floatAbove(T.best.t);  // smallest float >= best.t
    // the ray in FP32, rounded to nearest: |of - o| <= u |o|, invf = inv (1 + e), |e| <= u = 2^-24 (|inv| <= 1e25: T.fast)
    const float of[3] = {(float)T.o.x, (float)T.o.y, (float)T.o.z}, invf[3] = {(float)T.inv.x, (float)T.inv.y, (float)T.inv.z};
    const bool pos[3] = {T.inv.x >= 0.0, T.inv.y >= 0.0, T.inv.z >= 0.0};
    uint32_t near_key = kQMissKey, near_a = 0;
    auto push = [&](uint32_t key, uint32_t a) {
        if (T.sp < stk.max_depth) {
            SmStackEntry e;
            e.key = key;
            e.a = a;
            stk.put(T.sp++, e);
            T.top_key = key;  // (the stack's top, cached: travPopCached, mcrt_lanesm.hpp)
            T.top_a = a;
        } else {
            cnt.overflow = 1;
        }
    };
    uint32_t bi = T.node_a;
    bool more = true;
    while (more) {
        const QBlock b = qFetch(qv, bi++);
        const uint32_t n = (b.w[3] >> 24) & 0x7Fu;
        more = (b.w[3] >> 31) != 0u;
        float A[3], Cn[3], Cf[3];
        uint32_t wn[3], wf[3];  // cell coordinates of the planes the ray enters / leaves through, four children per word
        for (int ax = 0; ax < 3; ax++) {
            // t(q) = (origin + q cell - o) inv = q A + C with A = cell inv, C = (origin - o) inv, all in FP32: cell is a power of
            // two >= 2^-126 (quantiseAxis), so A = cell invf is exact and off by <= u |A|; d = fl(origin - of) is off the true
            // difference by <= u |d| + u |o|, so C = fl(d invf) by <= 3u |C| + 1.01u |o inv|; the fma rounds by <= u |t|. The
            // margin m = 8u (255 |A| + |C|) + 4u |of invf| covers the sum with room: the planes the ray enters through move
            // towards it, the others away, every slab only grows (by ~1e-7 of the ray's distance from the coordinate origin —
            // nothing next to the 1/255 quantisation of the boxes).
            const float cell = bitsFloat((((b.w[3] >> (8 * ax)) & 0xFFu) - 1u) << 23);  // 2^(e - 128)
            A[ax] = cell * invf[ax];
            const float C = (bitsFloat(b.w[ax]) - of[ax]) * invf[ax];
            const float m = fmaf(fmaf(255.0f, fabsf(A[ax]), fabsf(C)), 4.76837158203125e-07f, fmaf(fabsf(of[ax] * invf[ax]), 2.384185791015625e-07f, 1e-30f));
            Cn[ax] = C - m;
            Cf[ax] = C + m;
            wn[ax] = pos[ax] ? b.w[4 + 2 * ax] : b.w[5 + 2 * ax];
            wf[ax] = pos[ax] ? b.w[5 + 2 * ax] : b.w[4 + 2 * ax];
        }
        uint32_t key[4], a[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 4; c++) {
            const float tnx = fmaf((float)((wn[0] >> (8 * c)) & 0xFFu), A[0], Cn[0]);
            const float tny = fmaf((float)((wn[1] >> (8 * c)) & 0xFFu), A[1], Cn[1]);
            const float tnz = fmaf((float)((wn[2] >> (8 * c)) & 0xFFu), A[2], Cn[2]);
            const float tfx = fmaf((float)((wf[0] >> (8 * c)) & 0xFFu), A[0], Cf[0]);
            const float tfy = fmaf((float)((wf[1] >> (8 * c)) & 0xFFu), A[1], Cf[1]);
            const float tfz = fmaf((float)((wf[2] >> (8 * c)) & 0xFFu), A[2], Cf[2]);
            const float lo = fmaxf(fmaxf(tnx, tny), tnz), hi = fminf(fminf(tfx, tfy), tfz);
            const float t = fmaxf(lo, 0.0f);
            const bool keep = (uint32_t)c < n && hi >= t && t <= best_up;
            if (kCount) cnt.node_tests += (uint32_t)c < n ? 1u : 0u;
            const uint32_t m = (b.w[14 + c / 2] >> (16 * (c % 2))) & 0x1FFu;
            key[c] = keep ? ((floatBits(t) & ~0x1FFu) | m) : kQMissKey;
            a[c] = b.w[10 + c];
        }
        auto exchange = [&](int i, int j) {  // afterwards key[i] <= key[j]
            const bool sw = key[j] < key[i];
            const uint32_t ki = sw ? key[j] : key[i], kj = sw ? key[i] : key[j];
            const uint32_t ai = sw ? a[j] : a[i], aj = sw ? a[i] : a[j];
            key[i] = ki; key[j] = kj; a[i] = ai; a[j] = aj;
        };
        exchange(0, 1);
        exchange(2, 3);
        exchange(0, 2);
        exchange(1, 3);
        exchange(1, 2);
        if (key[3] != kQMissKey) push(key[3], a[3]);
        if (key[2] != kQMissKey) push(key[2], a[2]);
        if (key[1] != kQMissKey) push(key[1], a[1]);
        // the block's nearest against the nearest of the node's earlier blocks (nodes with more than 4 children)
        const bool better = key[0] < near_key;
        const uint32_t lose_key = better ? near_key : key[0], lose_a = better ? near_a : a[0];
        if (better) {
            near_key = key[0];
            near_a = a[0];
        }
        if (lose_key != kQMissKey) push(lose_key, lose_a);
    }
    if (near_key != kQMissKey) {
        T.node_a = near_a;
        T.node_m = near_key & 0x1FFu;
    } else if (kLazyPop) {  // (the trace kernel's deferred-leaf forms pop at ONE site per loop iteration: mcrt_kernels.hpp)
        T.active = false;
        T.need_pop = true;
    } else {
        travPop(T, stk);
    }
}

// ---- The lean visit (round 5) -------------------------------------------------------------------------------------------------
// What a visit of travInnerStepQ costs beyond its slab tests (ISA of wfTraceKernel<PoolRays, false, 3>, ~230 issue slots): ~25 slots
// of prologue that depend on the RAY only and were redone at every node (seven v_cvt_f32_f64 and three v_cmp_*_f64 at the FP64 rate:
// the ray in FP32, floatAbove(best.t), the direction signs); ~20 for the "node with more than four children" loop around the block
// (the next-block fetch, the merge of a block's nearest with the earlier blocks', a fourth push) that a quaternary tree never needs;
// and ~55 in the pushes: up to four of them, each `if (key != miss) if (sp < max) if (sp < lds_depth) LDS else memory`, three nested
// exec-mask regions that a wave of 50 rays always enters. Here
//   * the FP32 ray and floatAbove(best.t) are kept per ray (LeanRay, mcrt_lanesm.hpp: set when the ray is loaded, best_up again
//     after every leaf step; a stale LARGER best_up only keeps a child the pop would cull anyway);
//   * kSingle (the tree has no node with more than four children: HostLayout knows) visits exactly one block;
//   * the three pushes are ONE block: the sorted keys' misses are a suffix, so with cnt children to push the entry of rank j goes
//     to stack position sp + cnt - j and a miss to a position ABOVE the new top (junk there is never read) - three unconditional
//     ds_write_b64 when sp + 3 fits the lane's LDS rows; only lanes whose stack is about to leave LDS take the general pushes.
//     No overflow test: the stack is sized to the tree's own bound (HostLayout::stack_bound), no walk can exceed it.
// Same blocks, same tests, same keys, same order of the kept children: the walk visits exactly what travInnerStepQ visits.
// (Tried on top, round 5: the entry and exit distance of an axis as ONE v_pk_fma_f32 - 12 instructions fewer per visit and 0.6 % SLOWER
// on C3 and C4, profiles/r05_ab_trace_pk_stack.log: a packed FP32 operation takes the SIMD as long as the two it replaces. Removed.
// Also tried: asking for the first word of the block the ray continues with at the END of the visit that chose it, so that the block
// is on its way to the L2 while the wave runs its leaf step - one more load and register per visit, 1.0 % / 0.7 % SLOWER on C3 / C4,
// profiles/r05_ab_builds_touch_flags.log. Removed.)
template <bool kLds, bool kCount, bool kSingle>
MCRT_HD void travInnerStepQLean(const QView<kLds>& qv, Trav& T, const LeanRay& R, const SmStack& stk, TraceCounters& cnt) {
    const float best_up = R.best_up;
    const float of[3] = {R.of[0], R.of[1], R.of[2]}, invf[3] = {R.invf[0], R.invf[1], R.invf[2]};
    const bool pos[3] = {invf[0] >= 0.0f, invf[1] >= 0.0f, invf[2] >= 0.0f};  // (|inv| >= 1: the float keeps the sign, never zero)
    uint32_t near_key = kQMissKey, near_a = 0;
    // (the general push, taken only when a lane's stack leaves its LDS rows: it keeps the bound test - the stacks are sized to the tree's
    // own bound, HostLayout::stack_bound, so it never fires, but a wrong bound must end as "traversal stack overflow", not as a write
    // into another lane's spill slab; the fast path below, sp + 3 <= lds_depth, cannot leave the lane's rows and needs none)
    auto push = [&](uint32_t key, uint32_t a) {
        if (T.sp < stk.max_depth) {
            SmStackEntry e;
            e.key = key;
            e.a = a;
            stk.put(T.sp++, e);
            T.top_key = key;
            T.top_a = a;
        } else {
            cnt.overflow = 1;
        }
    };
    uint32_t bi = T.node_a;
    bool more = true;
    while (more) {
        const QBlock b = qFetch(qv, bi++);
        const uint32_t n = (b.w[3] >> 24) & 0x7Fu;
        more = !kSingle && (b.w[3] >> 31) != 0u;
        float A[3], Cn[3], Cf[3];
        uint32_t wn[3], wf[3];
        for (int ax = 0; ax < 3; ax++) {  // (the error bounds: travInnerStepQ)
            const float cell = bitsFloat((((b.w[3] >> (8 * ax)) & 0xFFu) - 1u) << 23);
            A[ax] = cell * invf[ax];
            const float C = (bitsFloat(b.w[ax]) - of[ax]) * invf[ax];
            const float m = fmaf(fmaf(255.0f, fabsf(A[ax]), fabsf(C)), 4.76837158203125e-07f, fmaf(fabsf(of[ax] * invf[ax]), 2.384185791015625e-07f, 1e-30f));
            Cn[ax] = C - m;
            Cf[ax] = C + m;
            wn[ax] = pos[ax] ? b.w[4 + 2 * ax] : b.w[5 + 2 * ax];
            wf[ax] = pos[ax] ? b.w[5 + 2 * ax] : b.w[4 + 2 * ax];
        }
        uint32_t key[4], a[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 4; c++) {
            const float tnx = fmaf((float)((wn[0] >> (8 * c)) & 0xFFu), A[0], Cn[0]);
            const float tny = fmaf((float)((wn[1] >> (8 * c)) & 0xFFu), A[1], Cn[1]);
            const float tnz = fmaf((float)((wn[2] >> (8 * c)) & 0xFFu), A[2], Cn[2]);
            const float tfx = fmaf((float)((wf[0] >> (8 * c)) & 0xFFu), A[0], Cf[0]);
            const float tfy = fmaf((float)((wf[1] >> (8 * c)) & 0xFFu), A[1], Cf[1]);
            const float tfz = fmaf((float)((wf[2] >> (8 * c)) & 0xFFu), A[2], Cf[2]);
            const float lo = fmaxf(fmaxf(tnx, tny), tnz), hi = fminf(fminf(tfx, tfy), tfz);
            const float t = fmaxf(lo, 0.0f);
            const bool keep = (uint32_t)c < n && hi >= t && t <= best_up;
            if (kCount) cnt.node_tests += (uint32_t)c < n ? 1u : 0u;
            const uint32_t m = (b.w[14 + c / 2] >> (16 * (c % 2))) & 0x1FFu;
            key[c] = keep ? ((floatBits(t) & ~0x1FFu) | m) : kQMissKey;
            a[c] = b.w[10 + c];
        }
        auto exchange = [&](int i, int j) {  // afterwards key[i] <= key[j]
            const bool sw = key[j] < key[i];
            const uint32_t ki = sw ? key[j] : key[i], kj = sw ? key[i] : key[j];
            const uint32_t ai = sw ? a[j] : a[i], aj = sw ? a[i] : a[j];
            key[i] = ki; key[j] = kj; a[i] = ai; a[j] = aj;
        };
        exchange(0, 1);
        exchange(2, 3);
        exchange(0, 2);
        exchange(1, 3);
        exchange(1, 2);
        // key[1..3] ascending, misses last: cnt of them are pushed, farthest first (rank 3 at the bottom, rank 1 on top)
        const uint32_t cnt3 = key[3] != kQMissKey ? 3u : key[2] != kQMissKey ? 2u : key[1] != kQMissKey ? 1u : 0u;
        if (T.sp + 3 <= stk.lds_depth) {
            const uint32_t p1 = cnt3 > 1u ? cnt3 - 1u : 0u;               // rank 1: the new top (cnt3 == 0: junk at sp)
            const uint32_t p2 = cnt3 >= 2u ? cnt3 - 2u : 1u;              // rank 2 (a miss: junk at sp + 1, above the top)
            const uint32_t p3 = cnt3 == 3u ? 0u : 2u;                     // rank 3 (a miss: junk at sp + 2)
            SmStackEntry e;
            e.key = key[3]; e.a = a[3];
            stk.lds[stackSlot(T.sp + (int)p3, stk.lds_stride)] = e;
            e.key = key[2]; e.a = a[2];
            stk.lds[stackSlot(T.sp + (int)p2, stk.lds_stride)] = e;
            e.key = key[1]; e.a = a[1];
            stk.lds[stackSlot(T.sp + (int)p1, stk.lds_stride)] = e;
            T.sp += (int)cnt3;
            if (cnt3 != 0u) {
                T.top_key = key[1];
                T.top_a = a[1];
            }
        } else {  // (this lane's stack is about to leave its LDS rows: the general pushes)
            if (key[3] != kQMissKey) push(key[3], a[3]);
            if (key[2] != kQMissKey) push(key[2], a[2]);
            if (key[1] != kQMissKey) push(key[1], a[1]);
        }
        if (kSingle) {
            near_key = key[0];
            near_a = a[0];
        } else {
            const bool better = key[0] < near_key;
            const uint32_t lose_key = better ? near_key : key[0], lose_a = better ? near_a : a[0];
            if (better) {
                near_key = key[0];
                near_a = a[0];
            }
            if (lose_key != kQMissKey) push(lose_key, lose_a);
        }
    }
    if (near_key != kQMissKey) {
        T.node_a = near_a;
        T.node_m = near_key & 0x1FFu;
    } else {  // (popped at the loop's one pop site: mcrt_kernels.hpp)
        T.active = false;
        T.need_pop = true;
    }
}

}  // namespace mcrt
