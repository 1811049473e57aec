// Eight-wide quantised nodes for trees that stay in HBM (round 3).
//
// Why. The closest hit is a minimum over exact FP64 primitive tests with the lowest-index tie rule, so it does not depend
// on the SHAPE of the tree the walk goes through — only on which leaves it reaches. The device layout therefore need not
// mirror the reference's hierarchy (binary, 4-ary or 8-ary, bvh.cpp:165-449). Two measurements say what it should look like:
//   * a scattered 64-byte block fetch moves a whole 128-byte line across the fabric (profiles/r03_traffic_calibration.json:
//     2.00 x the bytes asked for) — half of every fetch of the 4-wide blocks (mcrt_qbvh.hpp) is wasted;
//   * the 4-wide visit costs ~290 wave instructions of which ~110 are the 5-exchange sort of {distance, link} pairs and up to
//     four conditional pushes, each a nest of three branches (stack full? LDS or spill?) — executed at 30-50 % of the lanes.
// Here the children of up to two (binary trees: three) reference levels share ONE 128-byte node: a float origin, a
// power-of-two cell per axis, 8-bit cell coordinates of eight child boxes, the index of the first inner child (inner children
// are contiguous, in slot order) and the primitive ranges of the leaf children. The children sit in slots by OCTANT — slot bit
// `axis` set = the child lies on the upper side of the node's centre along that axis — so "nearest first" is a bit trick
// instead of a sort: with octneg = the ray's direction signs, the hit children are visited in ascending (slot XOR octneg). A
// visit keeps its hit children as ONE group {first inner child, hit bits, inner mask, smallest entry distance}; descending
// into one of them pushes what is left of the group as one 8-byte stack entry (at most one push and one pop per visit, no
// sort). Hit leaves form a second group that the lane works off before it descends further, so that their hits prune what
// follows. (The structure follows Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide
// BVHs", HPG 2017, restated for exact FP64 leaves and the conservative FP32 slab test of mcrt_qbvh.hpp.)
//
// Exactness: as for the 4-wide blocks. A decoded box contains the reference's box of that child (rounded outwards by the
// host, verified with the kernels' own decode); the FP32 slab test moves entry planes towards the ray and exit planes away
// by a margin that covers every rounding; a group popped from the stack is dropped only if its smallest entry distance
// (rounded DOWN to 16 bits) exceeds the current best t. A leaf the exact walk reaches is therefore reached here, the
// primitive tests are the unchanged FP64 ones, and the result is the same minimum. Rays with a zero direction component (or
// |1/d| > 1e25) walk the exact 64-byte records as before (Trav::fast == false).
#pragma once

#include "mcrt_lanesm.hpp"

namespace mcrt {

// 128 bytes:
//   w[0..2]    float origin x, y, z (<= every child's lower bound)
//   w[3]       ex | ey << 8 | ez << 16 | imask << 24: cell = 2^(e - 128) per axis; imask = slots that hold inner children
//   w[4..15]   cell coordinates: plane p = 2 axis + side (side 0 = lower, 1 = upper); w[4 + 2 p] = slots 0..3 (slot s in
//              byte s), w[5 + 2 p] = slots 4..7
//   w[16]      index of the first inner child's node; inner child in slot s = w[16] + popcount(imask & ((1 << s) - 1))
//   w[17]      valid mask: slots in use (inner or leaf)
//   w[18..19]  primitive counts of the leaf children, slot s in byte s of w[18 + s / 4] (0 for inner and empty slots)
//   w[20..27]  leaf child in slot s: first primitive
//   w[28..31]  unused
struct alignas(128) WNode {
    uint32_t w[32];
};

struct WView {
    const WNode* nodes;
};

// The part of a wide walk's state that does not fit Trav: the group of hit LEAF children still to be tested.
struct WLeaves {
    uint32_t node = 0;  // wide node the leaves belong to
    uint32_t bits = 0;  // hit leaf children left, bit (slot XOR octneg)
};

MCRT_HD uint32_t octNeg(const Trav& T) { return (T.d.x < 0.0 ? 1u : 0u) | (T.d.y < 0.0 ? 2u : 0u) | (T.d.z < 0.0 ? 4u : 0u); }

// bit p of the result = bit (p XOR o) of m (8 bits): slot order -> visiting order of a ray with direction signs o
MCRT_HD uint32_t octPermute(uint32_t m, uint32_t o) {
    if (o & 1u) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (o & 2u) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (o & 4u) m = ((m & 0x0Fu) << 4) | ((m >> 4) & 0x0Fu);
    return m;
}

// the same permutation on both bytes of a 16-bit pair of masks
MCRT_HD uint32_t octPermute16(uint32_t m, uint32_t o) {
    if (o & 1u) m = ((m & 0x5555u) << 1) | ((m >> 1) & 0x5555u);
    if (o & 2u) m = ((m & 0x3333u) << 2) | ((m >> 2) & 0x3333u);
    if (o & 4u) m = ((m & 0x0F0Fu) << 4) | ((m >> 4) & 0x0F0Fu);
    return m;
}

MCRT_HD uint32_t popCount(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(m);
#else
    return (uint32_t)__builtin_popcount(m);
#endif
}

// Fast rays keep the current group of hit INNER children in Trav: node_a = first inner child of the node they belong to,
// node_m = hit bits (visiting order) | imask << 8 | (float bits of the group's smallest entry distance, upper 16) — the
// stack entry format too. Trav::active = something is left to do (group, stack, leaves).
constexpr uint32_t kWGroupBits = 0xFFu;

MCRT_HD bool wideWork(const Trav& T, const WLeaves& Lv, const PendLeaf& P) {
    return (T.node_m & kWGroupBits) != 0u || T.sp > 0 || Lv.bits != 0u || P.n != 0u;
}

// Scene::intersect / BVH::intersect start (bvh.cpp:84-88): the root box is tested exactly (one record). A fast ray then
// starts with the group "the root node" (first child 0, no inner mask: child index 0 whatever the slot), or with the root's
// primitives when the root is a leaf.
template <bool kAll, bool kCount>
MCRT_HD void travBeginW(const SmSceneView<kAll>& sv, Trav& T, WLeaves& Lv, PendLeaf& P, d3 start, d3 direction, d3 inv_direction, bool shadow,
                        const ShadowQuery* sq, TraceCounters& cnt) {
    travBegin<kAll, kCount>(sv, T, start, direction, inv_direction, shadow, sq, cnt);
    Lv.bits = 0u;
    Lv.node = 0u;
    P.n = 0u;
    if (T.fast && T.active) {
        if (T.node_m & kSmInner) {
            T.node_a = 0u;
            T.node_m = 1u;
        } else {
            P.a = T.node_a;
            P.n = T.node_m;
            T.node_m = 0u;
        }
    }
}

// One step of a fast ray that has no leaves waiting: take the next hit inner child (of the current group, else of the
// newest group on the stack that can still matter), fetch its node and test its eight children.
template <bool kCount>
MCRT_HD void travWideStep(const WView& wv, Trav& T, WLeaves& Lv, const SmStack& stk, TraceCounters& cnt) {
    uint32_t bits = T.node_m & kWGroupBits;
    if (bits == 0u) {
        bool got = false;
        while (T.sp > 0) {
            const SmStackEntry e = stk.get(--T.sp);
            if ((double)bitsFloat(e.key & 0xFFFF0000u) <= T.best.t) {
                T.node_a = e.a;
                T.node_m = e.key;
                bits = e.key & kWGroupBits;
                got = true;
                break;
            }
        }
        if (!got) {
            T.node_m = 0u;
            T.active = false;
            return;
        }
    }
    const uint32_t oct = octNeg(T);
    const uint32_t p = lowestBit(bits);
    const uint32_t s = p ^ oct;
    const uint32_t imask_parent = (T.node_m >> 8) & 0xFFu;
    const uint32_t child = T.node_a + popCount(imask_parent & ((1u << s) - 1u));
    const uint32_t rest = bits & (bits - 1u);
    if (rest) {  // what is left of the group waits on the stack
        if (T.sp < stk.max_depth) {
            SmStackEntry e;
            e.key = (T.node_m & ~kWGroupBits) | rest;
            e.a = T.node_a;
            stk.put(T.sp++, e);
        } else {
            cnt.overflow = 1;
        }
    }
    const WNode* nd = wv.nodes + child;
    uint32_t w[28];
    for (int k = 0; k < 28; k++) w[k] = nd->w[k];

    const float best_up = floatAbove(T.best.t);  // smallest float >= best.t
    // the ray in FP32, rounded to nearest: |of - o| <= u |o|, invf = inv (1 + e), |e| <= u = 2^-24 (|inv| <= 1e25: T.fast)
    const float of[3] = {(float)T.o.x, (float)T.o.y, (float)T.o.z}, invf[3] = {(float)T.inv.x, (float)T.inv.y, (float)T.inv.z};
    float A[3], Cn[3], Cf[3];
    uint32_t wn[3][2], wf[3][2];  // cell coordinates of the planes the ray enters / leaves through
    for (int ax = 0; ax < 3; ax++) {
        // t(q) = q A + C with A = cell invf (exact product: cell is a power of two >= 2^-126), C = (origin - of) invf; the margin
        // m = 8u (255 |A| + |C|) + 4u |of invf| covers the conversions, the difference, the product and the multiply-add
        // (derivation: travInnerStepQ, mcrt_qbvh.hpp): entry planes move towards the ray, exit planes away.
        const float cell = bitsFloat((((w[3] >> (8 * ax)) & 0xFFu) - 1u) << 23);  // 2^(e - 128)
        A[ax] = cell * invf[ax];
        const float C = (bitsFloat(w[ax]) - of[ax]) * invf[ax];
        const float m = fmaf(fmaf(255.0f, fabsf(A[ax]), fabsf(C)), 4.76837158203125e-07f, fmaf(fabsf(of[ax] * invf[ax]), 2.384185791015625e-07f, 1e-30f));
        Cn[ax] = C - m;
        Cf[ax] = C + m;
        const bool neg = (oct >> ax) & 1u;
        const int lo = 4 + 4 * ax, hi = 6 + 4 * ax;
        wn[ax][0] = neg ? w[hi] : w[lo];
        wn[ax][1] = neg ? w[hi + 1] : w[lo + 1];
        wf[ax][0] = neg ? w[lo] : w[hi];
        wf[ax][1] = neg ? w[lo + 1] : w[hi + 1];
    }
    // Eight slab tests, branch-free: a slot's entry distance when it is kept, +inf otherwise. Empty slots carry inverted boxes
    // (they miss unless the margins exceed the cell grid) and are masked out once, after the loop.
    uint32_t hits = 0u;
    const uint32_t imask = w[3] >> 24, valid = w[17] & 0xFFu;
    float tk[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < 8; c++) {
        const int h = c >> 2, sh = 8 * (c & 3);
        const float tnx = fmaf((float)((wn[0][h] >> sh) & 0xFFu), A[0], Cn[0]);
        const float tny = fmaf((float)((wn[1][h] >> sh) & 0xFFu), A[1], Cn[1]);
        const float tnz = fmaf((float)((wn[2][h] >> sh) & 0xFFu), A[2], Cn[2]);
        const float tfx = fmaf((float)((wf[0][h] >> sh) & 0xFFu), A[0], Cf[0]);
        const float tfy = fmaf((float)((wf[1][h] >> sh) & 0xFFu), A[1], Cf[1]);
        const float tfz = fmaf((float)((wf[2][h] >> sh) & 0xFFu), A[2], Cf[2]);
        const float lo = fmaxf(fmaxf(tnx, tny), tnz), hi = fminf(fminf(tfx, tfy), tfz);
        const float t = fmaxf(lo, 0.0f);
        const bool keep = hi >= t && t <= best_up;
        hits |= keep ? (1u << c) : 0u;
        tk[c] = keep ? t : INFINITY;
    }
    hits &= valid;
    if (kCount) cnt.node_tests += popCount(valid);
    // the group's smallest entry distance (over every kept slot: a lower bound of the inner children's)
    const float tmin = fminf(fminf(fminf(tk[0], tk[1]), fminf(tk[2], tk[3])), fminf(fminf(tk[4], tk[5]), fminf(tk[6], tk[7])));
    const uint32_t both_p = octPermute16((hits & imask) | ((hits & ~imask) << 8), oct);
    const uint32_t inner_p = both_p & 0xFFu, leaf_p = both_p >> 8;
    T.node_a = w[16];
    T.node_m = inner_p | (imask << 8) | (floatBits(tmin) & 0xFFFF0000u);  // (no inner hit: bits == 0, the rest is not read)
    Lv.node = child;
    Lv.bits = leaf_p;
    T.active = inner_p != 0u || T.sp > 0 || leaf_p != 0u;
}

// The lane's pending leaf is exhausted and its leaf group is not: the next leaf of the group (its range is read from the node).
MCRT_HD void travWideNextLeaf(const WView& wv, const Trav& T, WLeaves& Lv, PendLeaf& P) {
    if (P.n != 0u || Lv.bits == 0u) return;
    const uint32_t p = lowestBit(Lv.bits);
    Lv.bits &= Lv.bits - 1u;
    const uint32_t s = p ^ octNeg(T);
    const WNode* nd = wv.nodes + Lv.node;
    P.a = nd->w[20 + s];
    P.n = (nd->w[18 + (s >> 2)] >> (8 * (s & 3))) & 0xFFu;
}

// After a leaf step: a decided shadow ray drops everything; otherwise `active` follows what is left.
MCRT_HD void travWideAfterLeaf(Trav& T, WLeaves& Lv, PendLeaf& P) {
    if (!T.active) {  // travPendStep: occluded for sure
        Lv.bits = 0u;
        T.node_m = 0u;
        P.n = 0u;
        return;
    }
    T.active = wideWork(T, Lv, P);
}

}  // namespace mcrt
