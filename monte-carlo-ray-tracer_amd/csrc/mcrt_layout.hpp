// Host-side conversion of the reference's flattened scene (include/mcrt.h descriptors) into the
// layout the gfx950 kernels read (mcrt_scene.hpp). Used by mcrt_upload_scene; host only.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_scene.hpp"
#include "mcrt_qbvh.hpp"
#include "mcrt_bvh_shared.hpp"  // surfaceBounds

namespace mcrt {

constexpr size_t kFlatCopyMax = 4096;  // surfaces up to which the kind-sorted flat copy is built (the flat loop itself: MCRT_FLAT_MAX, default 64)

struct HostLayout {
    std::vector<double> node_bounds;   // [n][6], breadth-first, children contiguous
    std::vector<NodeMeta> node_meta;   // [n]
    std::vector<double> prim;          // [n][kPrimStride]
    std::vector<double> normal;        // [n][3] Triangle::normal_
    std::vector<double> shade_rec;     // [n][16] what a hit needs, ONE 128-byte line per surface: the face normal, material | kind << 32 as the fourth
                                       // word's bits, the three vertex normals (interpolating triangles), padding (mcrt_shade.hpp: kSurfRecWords)
    bool any_vn = false;
    // kind-sorted copy for the flat (tiny-scene) loop: triangles first, then spheres
    std::vector<double> flat_prim;     // [n][kPrimStride]
    std::vector<uint32_t> flat_index;  // [n] sorted slot -> surface index
    uint32_t flat_tris = 0;
    // FP32 cull records of the flat loop (mcrt_scene.hpp "FP32 cull"); empty = no cull for this scene
    std::vector<float> flat_pre;
    uint32_t pre_tri_pairs = 0, pre_sph_pairs = 0;
    double pre_centre[3] = {0.0, 0.0, 0.0}, pre_bound = 0.0;
    std::vector<Node64> nodes64;       // [n] box + meta records, same order as node_bounds
    std::vector<QBlock> qblocks;       // quantised child blocks (mcrt_qbvh.hpp), breadth-first over the inner nodes
    uint32_t q_root_a = 0, q_root_m = 0;
    bool q_single = false;             // every inner node is ONE block (no node with more than four children): travInnerStepQLean<.., kSingle>
    // FP32 cull records of the primitives in BVH order, one 128-byte record per ALIGNED pair (2p, 2p + 1) (mcrt_lanesm.hpp "leaf cull")
    uint32_t num_quadric_surfaces = 0;
    std::vector<double> surf_v_patched;  // surf_v with quadric record addresses (only when the scene has quadrics)
    // Most traversal-stack entries any depth-first walk of nodes64 can hold at once: on the way down a visit continues with one child
    // and pushes at most the others, so standing at a node the stack holds at most the sum over its ancestors of (children - 1) - a
    // property of the tree, computed at upload. The reference's frontier is an unbounded heap (bvh.cpp:80-129); the kernels' stacks
    // are sized to THIS bound (mcrt_upload_scene: kMaxStackDepth or more), so no ray can overflow them.
    uint32_t stack_bound = 0;
};

// (nodes are breadth-first with parents in front of their children: one forward pass)
inline uint32_t stackBound(const std::vector<Node64>& nodes) {
    const size_t n = nodes.size();
    if (n == 0) return 0;
    std::vector<uint32_t> held(n, 0u);  // entries on the stack while the walk stands at node i
    uint32_t bound = 0;
    for (size_t i = 0; i < n; i++) {
        if (!(nodes[i].m & kSmInner)) continue;
        const uint32_t count = nodes[i].m & 0xFFu, first = nodes[i].a;
        const uint32_t below = held[i] + (count ? count - 1u : 0u);
        bound = bound > below ? bound : below;
        for (uint32_t c = 0; c < count && (size_t)first + c < n; c++) held[first + c] = below;
    }
    return bound;
}

// One axis of one block: origin (float, rounded down), cell exponent and the cell coordinates of every child,
// chosen so that qDecode(lower) <= lo and qDecode(upper) >= hi hold in the kernels' own arithmetic.
inline bool quantiseAxis(const double* lo, const double* hi, int n, float& origin, uint8_t& exp_out, uint8_t* qlo, uint8_t* qhi) {
    double lo_min = lo[0], hi_max = hi[0];
    for (int c = 1; c < n; c++) {
        lo_min = lo[c] < lo_min ? lo[c] : lo_min;
        hi_max = hi[c] > hi_max ? hi[c] : hi_max;
    }
    if (!(fabs(lo_min) <= 3.0e38) || !(fabs(hi_max) <= 3.0e38)) return false;  // not representable around a float origin
    float o = (float)lo_min;
    if ((double)o > lo_min) o = nextafterf(o, -INFINITY);
    origin = o;
    const double extent = hi_max - (double)o;
    constexpr int kMinExp = -126;  // smallest cell: a normal float (the FP32 block visit builds 2^e from the exponent byte)
    int e = kMinExp;
    if (extent > 0.0) {
        int ex;
        frexp(extent / 255.0, &ex);  // extent/255 = f * 2^ex, f in [0.5, 1)  ->  2^ex >= extent/255
        e = ex;
    }
    for (;; e++) {
        if (e < kMinExp) e = kMinExp;
        if (e > 255 - kQExpBias) return false;
        const double cell = qCell((uint32_t)(e + kQExpBias));
        bool ok = true;
        for (int c = 0; c < n && ok; c++) {
            double ql = floor((lo[c] - (double)o) / cell), qh = ceil((hi[c] - (double)o) / cell);
            if (ql < 0.0) ql = 0.0;
            if (ql > 255.0) ql = 255.0;
            if (qh < 0.0) qh = 0.0;
            while (ql > 0.0 && qDecode(o, (uint32_t)ql, cell) > lo[c]) ql -= 1.0;
            while (qh <= 255.0 && qDecode(o, (uint32_t)qh, cell) < hi[c]) qh += 1.0;
            if (qh > 255.0 || qDecode(o, (uint32_t)ql, cell) > lo[c]) {
                ok = false;
                break;
            }
            qlo[c] = (uint8_t)ql;
            qhi[c] = (uint8_t)qh;
        }
        if (ok) {
            exp_out = (uint8_t)(e + kQExpBias);
            return true;
        }
    }
}

// Blocks for every inner node of the breadth-first Node64 array, in node order (so the top of the tree is a
// prefix of the block array too).
inline int buildQBlocks(HostLayout& L, std::string& err) {
    const uint32_t n = (uint32_t)L.nodes64.size();
    L.qblocks.clear();
    L.q_root_a = L.q_root_m = 0;
    L.q_single = false;
    if (n == 0) return MCRT_OK;
    std::vector<uint32_t> first_block(n, 0);
    uint32_t total = 0;
    for (uint32_t i = 0; i < n; i++)
        if (L.nodes64[i].m & kSmInner) {
            first_block[i] = total;
            total += ((L.nodes64[i].m & 0xFFu) + 3u) / 4u;
        }
    L.qblocks.assign(total, QBlock{});
    {
        uint32_t inner = 0;
        for (uint32_t i = 0; i < n; i++) inner += (L.nodes64[i].m & kSmInner) ? 1u : 0u;
        L.q_single = inner != 0 && total == inner;
    }
    auto link = [&](uint32_t node, uint32_t& a, uint32_t& m) {
        m = L.nodes64[node].m;
        a = (m & kSmInner) ? first_block[node] : L.nodes64[node].a;
    };
    link(0, L.q_root_a, L.q_root_m);
    for (uint32_t i = 0; i < n; i++) {
        const Node64& nd = L.nodes64[i];
        if (!(nd.m & kSmInner)) continue;
        const uint32_t count = nd.m & 0xFFu, nb = (count + 3u) / 4u;
        if (count == 0) {
            err = "BVH inner node without children";
            return MCRT_ERR_INVALID;
        }
        for (uint32_t b = 0; b < nb; b++) {
            QBlock& q = L.qblocks[first_block[i] + b];
            const uint32_t c0 = nd.a + b * 4u, nc = std::min<uint32_t>(4u, count - b * 4u);
            uint8_t exps[3], qlo[3][4], qhi[3][4];
            for (int ax = 0; ax < 3; ax++) {
                double lo[4], hi[4];
                for (uint32_t c = 0; c < nc; c++) {
                    lo[c] = L.nodes64[c0 + c].b[ax];
                    hi[c] = L.nodes64[c0 + c].b[3 + ax];
                }
                float o;
                if (!quantiseAxis(lo, hi, (int)nc, o, exps[ax], qlo[ax], qhi[ax])) {
                    err = "BVH bounds cannot be quantised (non-finite or beyond float range)";
                    return MCRT_ERR_UNSUPPORTED;
                }
                memcpy(&q.w[ax], &o, 4);
            }
            q.w[3] = (uint32_t)exps[0] | ((uint32_t)exps[1] << 8) | ((uint32_t)exps[2] << 16) | ((nc | (b + 1 < nb ? 0x80u : 0u)) << 24);
            for (int ax = 0; ax < 3; ax++)
                for (uint32_t c = 0; c < nc; c++) {  // one word per (axis, side), child c in byte c
                    q.w[4 + 2 * ax] |= (uint32_t)qlo[ax][c] << (8 * c);
                    q.w[5 + 2 * ax] |= (uint32_t)qhi[ax][c] << (8 * c);
                }
            for (uint32_t c = 0; c < nc; c++) {
                uint32_t a, m;
                link(c0 + c, a, m);
                q.w[10 + c] = a;
                q.w[14 + c / 2] |= m << (16 * (c % 2));
            }
        }
    }
    return MCRT_OK;
}

// Reference LinearNode array (depth-first, sibling links, bvh/bvh.hpp:68-74) -> breadth-first order in
// which the children of a node are contiguous and the top of the tree is a prefix of the array.
inline int convertNodes(const mcrt_scene_desc* s, std::vector<double>& bounds, std::vector<NodeMeta>& meta, std::string& err) {
    const uint32_t n = s->num_nodes;
    bounds.assign((size_t)n * 6, 0.0);
    meta.assign(n, NodeMeta{0u, 0u});
    if (n == 0) return MCRT_OK;
    std::vector<uint32_t> order;  // new index -> old index
    order.reserve(n);
    std::vector<uint32_t> first_child_new(n, 0), child_count(n, 0);
    order.push_back(0);
    for (size_t head = 0; head < order.size(); head++) {
        const uint32_t old = order[head];
        if (s->node_num_surfaces[old] != 0) continue;  // leaf (bvh.cpp:92)
        // inner: children are old+1 and its next_sibling chain (bvh.cpp:110-119)
        uint32_t c = old + 1, cnt = 0;
        first_child_new[head] = (uint32_t)order.size();
        while (c != 0 && c < n) {
            order.push_back(c);
            cnt++;
            if (order.size() > n) {
                err = "BVH node links are cyclic";
                return MCRT_ERR_INVALID;
            }
            c = s->node_next_sibling[c];
        }
        child_count[head] = cnt;
    }
    if (order.size() != n) {
        err = "BVH has unreachable nodes";
        return MCRT_ERR_INVALID;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t old = order[i];
        memcpy(&bounds[(size_t)i * 6], s->node_bounds + (size_t)old * 6, 48);
        if (s->node_num_surfaces[old] != 0) {
            if ((uint64_t)s->node_start_surface[old] + s->node_num_surfaces[old] > s->num_surfaces) {
                err = "BVH leaf range exceeds the surface array";
                return MCRT_ERR_INVALID;
            }
            meta[i] = NodeMeta{s->node_start_surface[old], s->node_num_surfaces[old]};
        } else {
            meta[i] = NodeMeta{first_child_new[i], kInnerFlag | child_count[i]};
        }
    }
    return MCRT_OK;
}

// FP32 cull records for the kind-sorted flat copy (L.flat_prim / L.flat_tris must be built). Error bounds: u = 2^-24,
// every input (start - centre, direction, v0 - centre, E1, E2, sphere centre - centre) is rounded to float once, a ray
// starts within `bound` (per axis) of the centre and has a unit direction; with Tmax >= |start - v0| for any such ray
// the FP32 values differ from the exact ones by at most
//   |d det| <= 14 u |E1||E2|     |d uN| <= 16 u |E2| Tmax     |d vN| <= 17 u |E1| Tmax     |d tN| <= 17 u |E1||E2| Tmax
// and the products the cull compares by at most 31 u |E1||E2|^2 Tmax (A), 31 u |E1|^2|E2| Tmax (B), 31 u (|E1||E2|)^2 Tmax (C),
// 40 u |E1||E2| ((|E1| + |E2|) Tmax + |E1||E2|) (W). The thresholds below are four times those. Spheres: with M = bound +
// |centre offset|_inf, |d so_i| <= 2 u M and the discriminant / 4 = nb^2 - so2 + r^2 is off by at most
// 2 u nb^2 + (12 u + 7 u M / r) so2 + (2 u r^2 + 7 u r M)  (using M |so| <= (r M + so2 M / r) / 2); again times four.
inline void buildFlatCull(HostLayout& L, uint32_t ns) {
    L.flat_pre.clear();
    L.pre_tri_pairs = L.pre_sph_pairs = 0;
    const uint32_t nt = L.flat_tris, nsph = ns - nt;
    if (ns == 0) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](const double* p, double r) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a] - r);
            hi[a] = std::max(hi[a], p[a] + r);
        }
    };
    for (uint32_t i = 0; i < ns; i++) {
        const double* r = &L.flat_prim[(size_t)i * kPrimStride];
        if (i < nt) {
            const double v1[3] = {r[0] + r[3], r[1] + r[4], r[2] + r[5]}, v2[3] = {r[0] + r[6], r[1] + r[7], r[2] + r[8]};
            grow(r, 0.0);
            grow(v1, 0.0);
            grow(v2, 0.0);
        } else {
            grow(r, fabs(r[3]));
        }
    }
    double H = 0.0;
    for (int a = 0; a < 3; a++) {
        if (!(fabs(lo[a]) <= 1e7) || !(fabs(hi[a]) <= 1e7)) return;  // keeps every FP32 product far from overflow
        L.pre_centre[a] = 0.5 * (lo[a] + hi[a]);
        H = std::max(H, 0.5 * (hi[a] - lo[a]));
    }
    if (!(H > 0.0)) return;
    L.pre_bound = 4.0 * H;  // rays may start up to four half-extents from the centre (a camera outside the room)
    const double u = ldexp(1.0, -24), bound = L.pre_bound;
    auto up = [](double x) {  // smallest float >= x (thresholds and additive constants round away from "reject")
        float f = (float)x;
        if ((double)f < x) f = nextafterf(f, INFINITY);
        return f;
    };
    auto down = [](double x) {
        float f = (float)x;
        if ((double)f > x) f = nextafterf(f, -INFINITY);
        return f;
    };
    auto len = [](const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    L.pre_tri_pairs = (nt + 1) / 2;
    L.pre_sph_pairs = (nsph + 1) / 2;
    L.flat_pre.assign((size_t)L.pre_tri_pairs * kTriPairFloats + (size_t)L.pre_sph_pairs * kSphPairFloats, 0.0f);
    for (uint32_t i = 0; i < 2 * L.pre_tri_pairs; i++) {
        float* q = &L.flat_pre[(size_t)(i / 2) * kTriPairFloats + (i & 1)];  // component c at q[2 c]
        if (i >= nt) {  // padding slot of an odd count: always rejected (and masked off anyway)
            q[2 * 9] = -INFINITY;
            continue;
        }
        const double* r = &L.flat_prim[(size_t)i * kPrimStride];
        double a[3];
        for (int c = 0; c < 3; c++) a[c] = r[c] - L.pre_centre[c];
        for (int c = 0; c < 3; c++) q[2 * c] = (float)a[c];
        for (int c = 0; c < 6; c++) q[2 * (3 + c)] = (float)r[3 + c];
        const double e1 = len(r + 3), e2 = len(r + 6), tmax = sqrt(3.0) * bound + len(a);
        const double k = 4.0 * u;
        double eu = k * 31.0 * e1 * e2 * e2 * tmax, ev = k * 31.0 * e1 * e1 * e2 * tmax, et = k * 31.0 * e1 * e1 * e2 * e2 * tmax;
        double ew = k * 40.0 * e1 * e2 * ((e1 + e2) * tmax + e1 * e2);
        const bool degenerate = !(e1 > 1e-12) || !(e2 > 1e-12) || !(eu > 1e-30) || !(ev > 1e-30) || !(et > 1e-30) || !(ew > 1e-30);
        if (degenerate) eu = ev = et = ew = INFINITY;  // always a survivor
        q[2 * 9] = up(eu);
        q[2 * 10] = up(ev);
        q[2 * 11] = up(et);
        q[2 * 12] = up(ew);
    }
    float* sph = L.flat_pre.data() + (size_t)L.pre_tri_pairs * kTriPairFloats;
    for (uint32_t i = 0; i < 2 * L.pre_sph_pairs; i++) {
        float* q = &sph[(size_t)(i / 2) * kSphPairFloats + (i & 1)];
        if (i >= nsph) {  // padding: always rejected
            q[2 * 3] = -INFINITY;
            q[2 * 4] = 0.0f;
            continue;
        }
        const double* r = &L.flat_prim[(size_t)(nt + i) * kPrimStride];
        double c[3], cinf = 0.0;
        for (int a = 0; a < 3; a++) {
            c[a] = r[a] - L.pre_centre[a];
            cinf = std::max(cinf, fabs(c[a]));
            q[2 * a] = (float)c[a];
        }
        const double rad = fabs(r[3]), M = bound + cinf;
        const double beta = 4.0 * (12.0 * u + 7.0 * u * M / rad), gamma = 4.0 * (2.0 * u * rad * rad + 7.0 * u * rad * M);
        if (!(rad > 0.0) || !(beta < 0.25) || !(gamma > 1e-30)) {  // always a survivor
            q[2 * 3] = INFINITY;
            q[2 * 4] = 0.0f;
        } else {
            q[2 * 3] = up(rad * rad + gamma);
            q[2 * 4] = down(1.0 - beta);
        }
    }
}


// A tree over INDEX RANGES for scenes that come without a BVH (Scene::intersect then loops over every surface,
// scene.cpp:163-174): node = a run of consecutive surfaces in the scene's own order, split into up to four consecutive runs
// until four surfaces or fewer are left. Poor as a BVH (the order of a scene file says little about position), but it
// gives the wavefront pipeline's trace kernel something to walk without renumbering anything — only the pipeline uses it
// (reconstruction filters need its shade kernel); the megakernels keep testing every surface. Arrays in the reference's
// linear form (children of node i: i + 1 and its next_sibling chain, bvh.cpp:110-119).
inline void synthRangeTree(const mcrt_scene_desc* s, std::vector<double>& bounds, std::vector<uint32_t>& start, std::vector<uint32_t>& count,
                           std::vector<uint32_t>& next) {
    const uint32_t ns = s->num_surfaces;
    std::vector<double> sb((size_t)ns * 6);
    for (uint32_t i = 0; i < ns; i++) surfaceBounds(s->surf_kind[i], s->surf_v + (size_t)i * 9, s->quadrics, &sb[(size_t)i * 6]);
    struct Rec {
        static uint32_t build(uint32_t a, uint32_t b, const std::vector<double>& sb, std::vector<double>& bounds, std::vector<uint32_t>& start,
                              std::vector<uint32_t>& count, std::vector<uint32_t>& next) {
            const uint32_t me = (uint32_t)start.size();
            double bb[6] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
                            -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
            for (uint32_t i = a; i < b; i++)
                for (int c = 0; c < 3; c++) {
                    bb[c] = sb[(size_t)i * 6 + c] < bb[c] ? sb[(size_t)i * 6 + c] : bb[c];
                    bb[3 + c] = sb[(size_t)i * 6 + 3 + c] > bb[3 + c] ? sb[(size_t)i * 6 + 3 + c] : bb[3 + c];
                }
            bounds.insert(bounds.end(), bb, bb + 6);
            start.push_back(a);
            next.push_back(0u);
            if (b - a <= 4u) {
                count.push_back(b - a);
                return me;
            }
            count.push_back(0u);
            const uint32_t part = (b - a + 3u) / 4u;
            uint32_t prev = 0;
            for (uint32_t x = a; x < b; x += part) {
                const uint32_t child = build(x, x + part < b ? x + part : b, sb, bounds, start, count, next);
                if (prev) next[prev] = child;
                prev = child;
            }
            return me;
        }
    };
    bounds.clear();
    start.clear();
    count.clear();
    next.clear();
    if (ns) Rec::build(0u, ns, sb, bounds, start, count, next);
}

inline int buildLayout(const mcrt_scene_desc* s, HostLayout& L, std::string& err) {
    const size_t ns = s->num_surfaces;
    L.prim.assign(ns * kPrimStride, 0.0);
    L.normal.assign(ns * 3, 0.0);
    L.any_vn = false;
    L.num_quadric_surfaces = 0;
    L.surf_v_patched.clear();
    for (size_t i = 0; i < ns; i++) {
        if (s->surf_material[i] >= s->num_materials) {
            err = "surface material index out of range";
            return MCRT_ERR_INVALID;
        }
        double* r = &L.prim[i * kPrimStride];
        if (s->surf_kind[i] == MCRT_SURF_SPHERE) {
            memcpy(r, s->surf_v + i * 9, 4 * sizeof(double));  // origin, radius
            r[9] = 1.0;
        } else if (s->surf_kind[i] == MCRT_SURF_TRIANGLE) {
            memcpy(r, s->surf_v + i * 9, 3 * sizeof(double));      // v0
            memcpy(r + 3, s->surf_e + i * 9, 6 * sizeof(double));  // E1, E2
            memcpy(&L.normal[i * 3], s->surf_e + i * 9 + 6, 3 * sizeof(double));
            const bool interp = s->surf_interpolate[i] != 0;
            if (interp && !s->surf_vn) {
                err = "surface interpolates normals but surf_vn is NULL";
                return MCRT_ERR_INVALID;
            }
            L.any_vn = L.any_vn || interp;
            r[9] = interp ? 2.0 : 0.0;
        } else if (s->surf_kind[i] == MCRT_SURF_QUADRIC) {
            const double idx = s->surf_v[i * 9];
            if (!s->quadrics || !(idx >= 0.0) || idx >= (double)s->num_quadrics || idx != (double)(uint32_t)idx) {
                err = "quadric surface names a record outside the quadrics array";
                return MCRT_ERR_INVALID;
            }
            r[0] = idx;  // replaced by the record's address once its home is known (patchQuadricAddresses)
            r[9] = 3.0;
            L.num_quadric_surfaces++;
        } else {
            err = "unsupported surface kind";
            return MCRT_ERR_UNSUPPORTED;
        }
    }
    L.shade_rec.assign(ns * 16, 0.0);
    for (size_t i = 0; i < ns; i++) {
        memcpy(&L.shade_rec[i * 16], &L.normal[i * 3], 3 * sizeof(double));
        const unsigned long long w = (unsigned long long)s->surf_material[i] | ((unsigned long long)s->surf_kind[i] << 32);
        memcpy(&L.shade_rec[i * 16 + 3], &w, 8);
        if (s->surf_kind[i] == MCRT_SURF_TRIANGLE && s->surf_interpolate[i] && s->surf_vn) memcpy(&L.shade_rec[i * 16 + 4], s->surf_vn + i * 9, 9 * sizeof(double));
    }
    for (uint32_t i = 0; i < s->num_lights; i++)
        if (s->light_surface[i] >= s->num_surfaces) {
            err = "light surface index out of range";
            return MCRT_ERR_INVALID;
        }
    // (only scenes small enough to ever be staged whole get the flat copy and its cull records: for a multi-million-triangle
    // scene they would be hundreds of MB of host work and upload that no kernel reads)
    const bool flat_possible = ns <= kFlatCopyMax;
    L.flat_prim.assign(flat_possible ? ns * kPrimStride : 0, 0.0);
    L.flat_index.assign(flat_possible ? ns : 0, 0u);
    L.flat_tris = 0;
    size_t slot = 0;
    for (int pass = 0; pass < 2 && flat_possible; pass++)
        for (size_t i = 0; i < ns; i++) {
            const bool sphere = s->surf_kind[i] == MCRT_SURF_SPHERE;
            if (sphere != (pass == 1)) continue;
            memcpy(&L.flat_prim[slot * kPrimStride], &L.prim[i * kPrimStride], kPrimStride * sizeof(double));
            L.flat_index[slot] = (uint32_t)i;
            slot++;
            if (!sphere) L.flat_tris++;
        }
    if (flat_possible) buildFlatCull(L, (uint32_t)ns);
    else {
        L.flat_pre.clear();
        L.pre_tri_pairs = L.pre_sph_pairs = 0;
    }
    // node records and child blocks: of the scene's BVH, or — for a scene without one — of a tree over index ranges that only
    // the wavefront pipeline walks (L.node_bounds / L.node_meta stay empty: the megakernels test every surface)
    auto nodeRecords = [&](const mcrt_scene_desc* d, std::vector<double>& nb, std::vector<NodeMeta>& nm) -> int {
        if (int rc = convertNodes(d, nb, nm, err)) return rc;
        L.nodes64.assign(d->num_nodes, Node64{});
        for (uint32_t i = 0; i < d->num_nodes; i++) {
            Node64& n = L.nodes64[i];
            memcpy(n.b, &nb[(size_t)i * 6], 48);
            const NodeMeta& m = nm[i];
            n.a = m.a;
            if (m.b & kInnerFlag) {
                const uint32_t count = m.b & ~kInnerFlag;
                if (count > 255) {
                    err = "BVH node with more than 255 children";
                    return MCRT_ERR_UNSUPPORTED;
                }
                n.m = kSmInner | count;
            } else {
                if (m.b > 255) {
                    err = "BVH leaf with more than 255 primitives";
                    return MCRT_ERR_UNSUPPORTED;
                }
                n.m = m.b;
            }
            n.pad0 = n.pad1 = 0;
        }
        L.stack_bound = stackBound(L.nodes64);
        return buildQBlocks(L, err);
    };
    if (s->num_nodes) return nodeRecords(s, L.node_bounds, L.node_meta);
    L.node_bounds.clear();
    L.node_meta.clear();
    std::vector<double> tb;
    std::vector<uint32_t> ts, tc, tn;
    synthRangeTree(s, tb, ts, tc, tn);
    mcrt_scene_desc t = *s;
    t.num_nodes = (uint32_t)ts.size();
    t.node_bounds = tb.data();
    t.node_start_surface = ts.data();
    t.node_num_surfaces = tc.data();
    t.node_next_sibling = tn.data();
    std::vector<double> nb;
    std::vector<NodeMeta> nm;
    std::string ignored;
    std::swap(err, ignored);
    if (nodeRecords(&t, nb, nm) != MCRT_OK) {  // (surfaces without finite bounds: no tree, the pipeline stays closed to this scene)
        L.nodes64.clear();
        L.qblocks.clear();
        L.q_root_a = L.q_root_m = 0;
        L.stack_bound = 0;
    }
    std::swap(err, ignored);
    return MCRT_OK;
}

// Quadric surfaces: put the address of each one's record (`records` = where the [n][22] array lives for the code that
// will read it: device memory for the kernels, host memory for the CPU harness) into its primitive record and into a
// copy of surf_v.
inline void patchQuadricAddresses(const mcrt_scene_desc* s, HostLayout& L, const double* records) {
    if (L.num_quadric_surfaces == 0) return;
    L.surf_v_patched.assign(s->surf_v, s->surf_v + (size_t)s->num_surfaces * 9);
    for (size_t i = 0; i < s->num_surfaces; i++)
        if (s->surf_kind[i] == MCRT_SURF_QUADRIC) {
            const double slot = quadricSlot(records + (size_t)s->surf_v[i * 9] * 22);
            L.prim[i * kPrimStride] = slot;
            L.surf_v_patched[i * 9] = slot;
        }
}

}  // namespace mcrt
