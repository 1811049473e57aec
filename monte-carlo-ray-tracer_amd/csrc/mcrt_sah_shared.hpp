// The reference's binned-SAH builders (bvh/bvh.cpp:165-288 recursiveBuildBinarySAH, :290-426 recursiveBuildQuaternarySAH,
// :451-473 arbitrarySplit) as a LEVEL-SYNCHRONOUS build: all open nodes of a depth are processed together, so that the work
// per level is a handful of passes over the surfaces (data-parallel: the GPU path, mcrt_sah_gpu.hip) plus one small
// decision per open node. Shared by the GPU path, by its host twin (buildSahLevelsHost below: the same passes as loops,
// mcrt_bvh_build_sah_levels) and by the tests.
//
// Why the tree is the reference's, bit for bit: a node's split depends only on the SET of its surfaces and their order
// (centroid bounds, per-bin counts and boxes are minima, maxima and integer sums: order-free and exact; the cost of every
// split position is evaluated with the reference's expression in the reference's order; the first minimum wins), the
// partition keeps the surfaces' relative order, and arbitrarySplit deals them round-robin. None of that needs the
// recursion: a node's surfaces are a contiguous run of one working order, its children are sub-runs in child order, and
// the final order is the depth-first order of the leaves (BVH::compact, bvh.cpp:428-449).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "mcrt_bvh_shared.hpp"

namespace mcrt {

constexpr uint32_t kSahLeaf = 8, kSahMaxLeaf = 0xFF;  // BVH::leaf_surfaces, max_leaf_surfaces (bvh.hpp:91-92)
constexpr uint32_t kSahMaxBins = 16;                  // bins per axis the level-synchronous build supports (reference defaults: 16 / 8)

// BoundingBox as the builders use it (bounding-box.hpp:25-26, bounding-box.cpp:35-40,56-72)
struct SahBox {
    double mn[3], mx[3];
    MCRT_HD void reset() {
        for (int c = 0; c < 3; c++) {
            mn[c] = 1.7976931348623157e308;
            mx[c] = -1.7976931348623157e308;
        }
    }
    MCRT_HD void merge6(const double* b) {
        for (int c = 0; c < 3; c++) {
            if (mn[c] > b[c]) mn[c] = b[c];
            if (mx[c] < b[3 + c]) mx[c] = b[3 + c];
        }
    }
    MCRT_HD void mergePoint(const double* p) {
        for (int c = 0; c < 3; c++) {
            if (mn[c] > p[c]) mn[c] = p[c];
            if (mx[c] < p[c]) mx[c] = p[c];
        }
    }
    MCRT_HD double area() const {
        for (int c = 0; c < 3; c++)
            if (mn[c] > mx[c]) return 0.0;
        const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.0 * (dx * dy + dx * dz + dy * dz);
    }
    MCRT_HD void store(double* b) const {
        for (int c = 0; c < 3; c++) {
            b[c] = mn[c];
            b[3 + c] = mx[c];
        }
    }
};

struct SahSeg {  // an open node: `size` (> 8) surfaces at `start` of the working order
    uint32_t start, size, node;
    uint32_t rule;  // 4: quaternary rule; 2: binary rule (also every node below a quaternary node that fell back to it, bvh.cpp:313-318)
    double box[6];  // the node's box (root: the scene box, bvh.cpp:20; else the union of its surfaces' boxes)
};
enum : uint32_t { kSahLeafMode = 0, kSahBinary = 1, kSahQuad = 2, kSahArb = 3 };
struct SahPlan {  // from the centroid bounds: how the node's surfaces are binned
    uint32_t mode;  // kSahLeafMode: no usable axis and <= 255 surfaces, stays a leaf; kSahArb: no usable axis, dealt round-robin
    uint32_t arb;   // kSahArb: parts
    int ax, ay;     // binned axes (kSahBinary: ax only)
    double mn[2], dim[2];
};
struct SahSplit {  // from the bins: the split, or none
    uint32_t mode, arb;
    int split_x, split_y;
    uint32_t child_rule;
    uint32_t child_size[4];  // kSahBinary: children 0, 1; kSahQuad: child (ix > split_x) | (iy > split_y) << 1; kSahArb: sizes of the parts
    double child_box[4][6];  // (not for kSahArb: the parts' boxes are gathered separately)
};

// quaternary(): :296-318; binary(): :170-185
MCRT_HD void sahPlan(const SahSeg& seg, const double* ce /* centroid bounds mn[3], mx[3] */, SahPlan& p) {
    const double dims[3] = {ce[3] - ce[0], ce[4] - ce[1], ce[5] - ce[2]};
    p.arb = 0;
    p.ay = 0;
    p.mn[1] = p.dim[1] = 0.0;
    if (seg.rule == 4) {
        int ax, ay;
        if (dims[0] > dims[1]) {
            ax = 0;
            ay = dims[1] > dims[2] ? 1 : 2;
        } else {
            if (dims[0] > dims[2]) { ax = 0; ay = 1; } else { ax = 1; ay = 2; }
        }
        if (!(dims[ax] < 1e-9 || dims[ay] < 1e-9)) {
            p.mode = kSahQuad;
            p.ax = ax;
            p.ay = ay;
            p.mn[0] = ce[ax];
            p.mn[1] = ce[ay];
            p.dim[0] = dims[ax];
            p.dim[1] = dims[ay];
            return;
        }
        // one usable axis: the binary rule on this node, and below it
    }
    const int axis = dims[0] > dims[1] ? (dims[0] > dims[2] ? 0 : 2) : (dims[1] > dims[2] ? 1 : 2);
    p.ax = axis;
    p.mn[0] = ce[axis];
    p.dim[0] = dims[axis];
    if (dims[axis] < 1e-9) {
        if (seg.size > kSahMaxLeaf) {
            p.mode = kSahArb;
            p.arb = 2;
        } else {
            p.mode = kSahLeafMode;
        }
        return;
    }
    p.mode = kSahBinary;
}

// getIdx, :186-190 / :319-325. Returns the cell (kSahBinary: the bin; kSahQuad: ix * bins + iy).
MCRT_HD uint32_t sahCell(const SahPlan& p, const double* c /* centroid */, int bins) {
    const double fx = (c[p.ax] - p.mn[0]) / p.dim[0];
    int ix = (int)floor(fx * (double)bins);
    if (ix > bins - 1) ix = bins - 1;
    if (p.mode != kSahQuad) return (uint32_t)ix;
    const double fy = (c[p.ay] - p.mn[1]) / p.dim[1];
    int iy = (int)floor(fy * (double)bins);
    if (iy > bins - 1) iy = bins - 1;
    return (uint32_t)(ix * bins + iy);
}

// :199-237 (binary), :333-388 (quaternary): the split positions in the reference's order, first minimum wins.
// count / bbox: the node's bins ([cells], [cells][6] as mn[3], mx[3]; an empty bin holds the empty box).
MCRT_HD void sahEvaluate(const SahSeg& seg, const SahPlan& p, int bins, const uint32_t* count, const double* bbox, SahSplit& s) {
    s.mode = p.mode;
    s.arb = p.arb;
    s.split_x = s.split_y = 0;
    s.child_rule = 2;
    for (int k = 0; k < 4; k++) s.child_size[k] = 0;
    if (p.mode == kSahLeafMode) return;
    if (p.mode == kSahArb) {
        for (uint32_t k = 0; k < p.arb; k++) s.child_size[k] = (seg.size - k + p.arb - 1u) / p.arb;
        return;
    }
    SahBox node;
    for (int c = 0; c < 3; c++) {
        node.mn[c] = seg.box[c];
        node.mx[c] = seg.box[3 + c];
    }
    const double node_area = node.area();
    double min_cost = 1.7976931348623157e308;
    if (p.mode == kSahBinary) {
        int split_bin = 0;
        for (int i = 0; i + 1 < bins; i++) {
            unsigned long long a_count = 0, b_count = 0;
            SahBox a_bb, b_bb;
            a_bb.reset();
            b_bb.reset();
            for (int j = 0; j < i + 1; j++) {
                a_count += count[j];
                a_bb.merge6(bbox + 6 * j);
            }
            for (int j = i + 1; j < bins; j++) {
                b_count += count[j];
                b_bb.merge6(bbox + 6 * j);
            }
            const double cost = 1.0 + ((double)a_count * a_bb.area() + (double)b_count * b_bb.area()) / node_area;
            if (cost < min_cost) {
                split_bin = i;
                min_cost = cost;
            }
        }
        if (min_cost > (double)seg.size) {
            if (seg.size > kSahMaxLeaf) {
                s.mode = kSahArb;
                s.arb = 2;
                for (uint32_t k = 0; k < 2; k++) s.child_size[k] = (seg.size - k + 1u) / 2u;
            } else {
                s.mode = kSahLeafMode;
            }
            return;
        }
        s.split_x = split_bin;
        SahBox cb[2];
        cb[0].reset();
        cb[1].reset();
        for (int j = 0; j < bins; j++) {
            const int k = j <= split_bin ? 0 : 1;
            s.child_size[k] += count[j];
            if (count[j]) cb[k].merge6(bbox + 6 * j);
        }
        cb[0].store(s.child_box[0]);
        cb[1].store(s.child_box[1]);
        return;
    }
    int split_x = 0, split_y = 0;
    for (int i = 0; i < bins - 1; i++)
        for (int j = 0; j < bins - 1; j++) {
            double cost = 0.0;
            for (int v = 0; v < 4; v++) {
                const int x0 = (v & 1) ? i + 1 : 0, x1 = (v & 1) ? bins : i + 1;
                const int y0 = (v & 2) ? j + 1 : 0, y1 = (v & 2) ? bins : j + 1;
                unsigned long long n = 0;
                SahBox q;
                q.reset();
                for (int x = x0; x < x1; x++)
                    for (int y = y0; y < y1; y++) {
                        n += count[x * bins + y];
                        q.merge6(bbox + 6 * (x * bins + y));
                    }
                cost += q.area() * (double)n;
            }
            cost = 1.0 + cost / node_area;
            if (cost < min_cost) {
                split_x = i;
                split_y = j;
                min_cost = cost;
            }
        }
    if (min_cost > (double)seg.size) {
        if (seg.size > kSahMaxLeaf) {
            s.mode = kSahArb;
            s.arb = 4;
            s.child_rule = 4;
            for (uint32_t k = 0; k < 4; k++) s.child_size[k] = (seg.size - k + 3u) / 4u;
        } else {
            s.mode = kSahLeafMode;
        }
        return;
    }
    s.split_x = split_x;
    s.split_y = split_y;
    s.child_rule = 4;
    SahBox cb[4];
    for (int k = 0; k < 4; k++) cb[k].reset();
    for (int x = 0; x < bins; x++)
        for (int y = 0; y < bins; y++) {
            const int k = (x > split_x ? 1 : 0) | (y > split_y ? 2 : 0);
            s.child_size[k] += count[x * bins + y];
            if (count[x * bins + y]) cb[k].merge6(bbox + 6 * (x * bins + y));
        }
    for (int k = 0; k < 4; k++) cb[k].store(s.child_box[k]);
}

// The child a surface of a split node goes to, from its cell (:239-247 / :398-409).
MCRT_HD uint32_t sahChildOfCell(const SahSplit& s, uint32_t cell, int bins) {
    if (s.mode == kSahBinary) return (int)cell <= s.split_x ? 0u : 1u;
    const int ix = (int)cell / bins, iy = (int)cell % bins;
    return (ix > s.split_x ? 1u : 0u) | (iy > s.split_y ? 2u : 0u);
}

// ---- the node table both drivers fill, and its depth-first numbering (BVH::compact, bvh.cpp:428-449)
struct SahTree {
    std::vector<double> box;                          // [nodes][6]
    std::vector<uint32_t> first_child, child_count;   // children are consecutive table entries
    std::vector<uint32_t> start, count;               // run of the working order; count != 0: leaf
    uint32_t add(const double* b, uint32_t start_, uint32_t leaf_count) {
        const uint32_t id = (uint32_t)start.size();
        box.insert(box.end(), b, b + 6);
        first_child.push_back(0);
        child_count.push_back(0);
        start.push_back(start_);
        count.push_back(leaf_count);
        return id;
    }
};

// Open nodes of the next level from the splits of this one. `arb_box`: the boxes of the round-robin parts of kSahArb
// nodes, in the order the parts appear ([part][6]); consumed front to back.
inline void sahGrow(SahTree& T, const std::vector<SahSeg>& segs, const std::vector<SahSplit>& splits, const double* arb_box,
                    std::vector<SahSeg>& next) {
    next.clear();
    size_t arb_at = 0;
    for (size_t i = 0; i < segs.size(); i++) {
        const SahSeg& g = segs[i];
        const SahSplit& s = splits[i];
        if (s.mode == kSahLeafMode) {  // :181-184 / :230-236: no split and it fits a leaf
            T.count[g.node] = g.size;
            continue;
        }
        const uint32_t parts = s.mode == kSahBinary ? 2u : s.mode == kSahQuad ? 4u : s.arb;
        uint32_t at = g.start;
        T.first_child[g.node] = (uint32_t)T.start.size();
        uint32_t made = 0;
        for (uint32_t k = 0; k < parts; k++) {
            const uint32_t sz = s.child_size[k];
            const double* b = s.mode == kSahArb ? arb_box + 6 * (arb_at++) : s.child_box[k];
            if (sz == 0) continue;  // (an arbitrary split of > 255 surfaces has no empty part)
            const uint32_t id = T.add(b, at, sz <= kSahLeaf ? sz : 0u);
            made++;
            if (sz > kSahLeaf) {
                SahSeg c;
                c.start = at;
                c.size = sz;
                c.node = id;
                c.rule = s.child_rule;
                memcpy(c.box, b, 48);
                next.push_back(c);
            }
            at += sz;
        }
        T.child_count[g.node] = made;
    }
}

inline void sahCompactInto(const SahTree& T, uint32_t node, uint32_t next_sibling, const uint32_t* subtree, mcrt_bvh* B) {
    // iterative depth-first walk (trees over coincident centroids get deep)
    struct Item { uint32_t node, next; };
    std::vector<Item> stack;
    stack.push_back(Item{node, next_sibling});
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        B->start.push_back(T.start[it.node]);
        B->count.push_back((uint32_t)(uint8_t)T.count[it.node]);  // LinearNode::num_surfaces is a uint8_t (bvh.cpp:433)
        B->next.push_back(it.next);
        for (int c = 0; c < 6; c++) B->bounds.push_back(T.box[(size_t)it.node * 6 + c]);
        const uint32_t nc = T.child_count[it.node], fc = T.first_child[it.node];
        // children in order: push in reverse; a child's id = parent's id + 1 + the subtree sizes of its elder siblings
        uint32_t my_id = (uint32_t)B->start.size() - 1u;
        std::vector<Item> kids(nc);
        uint32_t id = my_id + 1u;
        for (uint32_t k = 0; k < nc; k++) {
            const uint32_t after = id + subtree[fc + k];
            kids[k] = Item{fc + k, k + 1 < nc ? after : 0u};  // next sibling: right after the child's subtree; 0 = none (bvh.cpp:443-447)
            id = after;
        }
        for (uint32_t k = nc; k-- > 0;) stack.push_back(kids[k]);
    }
}

inline void sahFinish(const SahTree& T, const uint32_t* order, uint64_t n, mcrt_bvh* B) {
    const size_t nodes = T.start.size();
    std::vector<uint32_t> subtree(nodes, 1u);
    for (size_t i = nodes; i-- > 0;)  // children have larger table indices than their parent
        for (uint32_t k = 0; k < T.child_count[i]; k++) subtree[i] += subtree[T.first_child[i] + k];
    B->start.clear();
    B->count.clear();
    B->next.clear();
    B->bounds.clear();
    sahCompactInto(T, 0u, 0u, subtree.data(), B);
    B->order.assign(order, order + n);
    finishBvhDesc(B);
}

// The host twin of the GPU path: the same level loop with the per-surface passes as plain loops.
inline int buildSahLevelsHost(const double* bb, const double* centroid, uint64_t n, const double* scene_min, const double* scene_max, int arity,
                              int bins, mcrt_bvh* B) {
    std::vector<uint32_t> order(n), scratch(n);
    for (uint64_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    SahTree T;
    double root_box[6];
    for (int c = 0; c < 3; c++) {
        root_box[c] = scene_min[c];
        root_box[3 + c] = scene_max[c];
    }
    T.add(root_box, 0u, n <= kSahLeaf ? (uint32_t)n : 0u);
    std::vector<SahSeg> segs, next;
    if (n > kSahLeaf) {
        SahSeg r;
        r.start = 0;
        r.size = (uint32_t)n;
        r.node = 0;
        r.rule = (uint32_t)arity;
        memcpy(r.box, root_box, 48);
        segs.push_back(r);
    }
    const int cells_max = bins * bins;
    std::vector<SahPlan> plans;
    std::vector<SahSplit> splits;
    std::vector<uint32_t> count, cell;
    std::vector<double> bbox, arb_box;
    while (!segs.empty()) {
        const size_t S = segs.size();
        plans.resize(S);
        splits.resize(S);
        count.assign(S * cells_max, 0u);
        bbox.resize(S * cells_max * 6);
        arb_box.clear();
        for (size_t i = 0; i < S; i++) {
            const SahSeg& g = segs[i];
            SahBox ce;
            ce.reset();
            for (uint32_t p = g.start; p < g.start + g.size; p++) ce.mergePoint(centroid + (size_t)order[p] * 3);
            double ceb[6];
            ce.store(ceb);
            sahPlan(g, ceb, plans[i]);
            uint32_t* cnt = &count[i * cells_max];
            double* bx = &bbox[i * cells_max * 6];
            for (int c = 0; c < cells_max; c++) {
                SahBox e;
                e.reset();
                e.store(bx + 6 * c);
            }
            if (plans[i].mode == kSahBinary || plans[i].mode == kSahQuad) {
                cell.resize(g.size);
                for (uint32_t p = 0; p < g.size; p++) {
                    const uint32_t s = order[g.start + p];
                    const uint32_t c = sahCell(plans[i], centroid + (size_t)s * 3, bins);
                    cell[p] = c;
                    cnt[c]++;
                    SahBox b;
                    for (int k = 0; k < 3; k++) {
                        b.mn[k] = bx[6 * c + k];
                        b.mx[k] = bx[6 * c + 3 + k];
                    }
                    b.merge6(bb + (size_t)s * 6);
                    b.store(bx + 6 * c);
                }
            }
            sahEvaluate(g, plans[i], bins, cnt, bx, splits[i]);
            const SahSplit& sp = splits[i];
            if (sp.mode == kSahBinary || sp.mode == kSahQuad) {  // order-preserving partition
                uint32_t at[4], run = g.start;
                for (int k = 0; k < 4; k++) {
                    at[k] = run;
                    run += sp.child_size[k];
                }
                for (uint32_t p = 0; p < g.size; p++) scratch[at[sahChildOfCell(sp, cell[p], bins)]++] = order[g.start + p];
                memcpy(&order[g.start], &scratch[g.start], (size_t)g.size * 4);
            } else if (sp.mode == kSahArb) {  // arbitrarySplit: surface i of the node goes to part i % N
                uint32_t at[4], run = g.start;
                for (uint32_t k = 0; k < sp.arb; k++) {
                    at[k] = run;
                    run += sp.child_size[k];
                }
                SahBox pb[4];
                for (int k = 0; k < 4; k++) pb[k].reset();
                for (uint32_t p = 0; p < g.size; p++) {
                    const uint32_t k = p % sp.arb;
                    scratch[at[k]++] = order[g.start + p];
                    pb[k].merge6(bb + (size_t)order[g.start + p] * 6);
                }
                memcpy(&order[g.start], &scratch[g.start], (size_t)g.size * 4);
                for (uint32_t k = 0; k < sp.arb; k++) {
                    double b6[6];
                    pb[k].store(b6);
                    arb_box.insert(arb_box.end(), b6, b6 + 6);
                }
            }
        }
        sahGrow(T, segs, splits, arb_box.data(), next);
        segs.swap(next);
    }
    sahFinish(T, order.data(), n, B);
    return MCRT_OK;
}

}  // namespace mcrt
