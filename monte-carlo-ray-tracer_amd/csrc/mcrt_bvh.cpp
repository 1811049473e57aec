// Octree BVH builder, host path and the scene re-ordering helper (include/mcrt.h: mcrt_bvh_build_octree,
// mcrt_scene_with_bvh). Algorithm and its equivalence with the reference's builder: mcrt_bvh_shared.hpp. The GPU path
// (boxes, codes, radix sort on the device) is bvhOctreeGpu in mcrt_octree_gpu.hip.
#include <atomic>
#include <cmath>
#include <memory>
#include <numeric>
#include <thread>

#include "mcrt_bvh_shared.hpp"
#include "mcrt_sah_shared.hpp"
#include "mcrt_internal.hpp"

using namespace mcrt;

struct mcrt_scene {  // owns every array its descriptor points into
    std::vector<uint8_t> kind, interp;
    std::vector<uint32_t> material, light_surface, node_start, node_count, node_next;
    std::vector<double> area, v, e, vn, node_bounds, light_cdf, quadrics;
    std::vector<mcrt_material> materials;
    mcrt_scene_desc desc;
};

namespace {

bool sceneUsable(const mcrt_scene_desc* s) {
    if (!s || s->abi_version != MCRT_ABI_VERSION || s->num_surfaces == 0 || !s->surf_kind || !s->surf_v) return false;
    for (uint32_t i = 0; i < s->num_surfaces; i++)
        if (s->surf_kind[i] > MCRT_SURF_QUADRIC || (s->surf_kind[i] == MCRT_SURF_QUADRIC && (!s->quadrics || s->surf_v[9 * (size_t)i] >= s->num_quadrics)))
            return false;
    return true;
}

int buildHost(const mcrt_scene_desc* s, mcrt_bvh* B) {
    const uint64_t n = s->num_surfaces;
    double mn[3], mx[3];
    bvhRootCube(s, mn, mx);
    std::vector<double> bb(n * 6);
    std::vector<unsigned long long> code(n);
    for (uint64_t i = 0; i < n; i++) {
        double* b = &bb[i * 6];
        surfaceBounds(s->surf_kind[i], s->surf_v + 9 * i, s->quadrics, b);
        // SurfaceCentroid = BB().centroid() = (max + min) / 2 (bvh.cpp:475-476, bounding-box.cpp:30-33)
        code[i] = cellCode((b[3] + b[0]) / 2.0, (b[4] + b[1]) / 2.0, (b[5] + b[2]) / 2.0, mn, mx);
    }
    std::vector<uint32_t> index(n);
    std::iota(index.begin(), index.end(), 0u);
    std::stable_sort(index.begin(), index.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    std::vector<unsigned long long> keys(n);
    std::vector<double> sorted_bb(n * 6);
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = code[index[i]];
        memcpy(&sorted_bb[i * 6], &bb[(size_t)index[i] * 6], 48);
    }
    return assembleOctreeBvh(keys.data(), index.data(), sorted_bb.data(), n, B) ? MCRT_OK : MCRT_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------
// The reference's binned-SAH builders, restated (bvh/bvh.cpp:165-288 binary, :290-426 quaternary, :451-473
// arbitrarySplit, :428-449 compact) with the same arithmetic and the same tie rules, so that the tree is the
// reference's: top-down, per node a histogram of the surfaces' box centroids over `bins` bins (one axis) or
// bins x bins (two axes), every split position costed with the surface-area heuristic against the node's box, an
// order-preserving partition. What differs is the execution: subtrees are independent once their surface lists exist,
// so they are built by as many host threads as there are, and the depth-first numbering is a second pass.
// ------------------------------------------------------------------------------------------------
struct Box6 {
    double mn[3], mx[3];
    Box6() {  // BoundingBox(), bounding-box.hpp:25-26
        for (int c = 0; c < 3; c++) {
            mn[c] = 1.7976931348623157e308;
            mx[c] = -1.7976931348623157e308;
        }
    }
    void merge(const double* b) {  // bounding-box.cpp:56-63
        for (int c = 0; c < 3; c++) {
            if (mn[c] > b[c]) mn[c] = b[c];
            if (mx[c] < b[3 + c]) mx[c] = b[3 + c];
        }
    }
    void merge(const Box6& o) {
        for (int c = 0; c < 3; c++) {
            if (mn[c] > o.mn[c]) mn[c] = o.mn[c];
            if (mx[c] < o.mx[c]) mx[c] = o.mx[c];
        }
    }
    void mergePoint(const double* p) {  // bounding-box.cpp:65-72
        for (int c = 0; c < 3; c++) {
            if (mn[c] > p[c]) mn[c] = p[c];
            if (mx[c] < p[c]) mx[c] = p[c];
        }
    }
    double area() const {  // bounding-box.cpp:35-40
        for (int c = 0; c < 3; c++)
            if (mn[c] > mx[c]) return 0.0;
        const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.0 * (dx * dy + dx * dz + dy * dz);
    }
};

struct SahNode {
    Box6 bb;
    std::vector<uint32_t> surfaces;
    std::vector<std::unique_ptr<SahNode>> children;
};

struct SahBuilder {
    const double* bb;        // [n][6] surface boxes
    const double* centroid;  // [n][3]
    int bins;
    std::atomic<int> free_threads;
    static constexpr size_t kLeaf = 8, kMaxLeaf = 0xFF;  // BVH::leaf_surfaces, max_leaf_surfaces (bvh.hpp:91-92)
    static constexpr size_t kSpawn = 20000;              // subtrees smaller than this stay on the current thread

    void arbitrarySplit(SahNode* node, size_t N) {  // bvh.cpp:451-473
        auto& S = node->surfaces;
        N = std::min(N, S.size());
        for (size_t i = 0; i < N; i++) node->children.emplace_back(new SahNode());
        for (size_t i = 0; i < S.size(); i++) {
            SahNode* c = node->children[i % N].get();
            c->surfaces.push_back(S[i]);
            c->bb.merge(bb + (size_t)S[i] * 6);
        }
        S.clear();
    }

    template <class F>
    void forChildren(SahNode* node, F recurse) {
        std::vector<std::thread> spawned;
        for (auto& c : node->children) {
            SahNode* child = c.get();
            if (child->surfaces.size() >= kSpawn && free_threads.fetch_sub(1) > 0) {
                spawned.emplace_back([this, child, recurse]() {
                    recurse(child);
                    free_threads.fetch_add(1);
                });
            } else {
                if (child->surfaces.size() >= kSpawn) free_threads.fetch_add(1);  // undo the failed claim
                recurse(child);
            }
        }
        for (auto& t : spawned) t.join();
    }

    void binary(SahNode* node) {  // recursiveBuildBinarySAH, bvh.cpp:165-288
        auto& S = node->surfaces;
        if (S.size() <= kLeaf) return;
        Box6 ce;
        for (uint32_t s : S) ce.mergePoint(centroid + (size_t)s * 3);
        const double dims[3] = {ce.mx[0] - ce.mn[0], ce.mx[1] - ce.mn[1], ce.mx[2] - ce.mn[2]};
        const int axis = dims[0] > dims[1] ? (dims[0] > dims[2] ? 0 : 2) : (dims[1] > dims[2] ? 1 : 2);
        auto recurse = [this](SahNode* n) { binary(n); };
        if (dims[axis] < 1e-9) {
            if (S.size() > kMaxLeaf) {
                arbitrarySplit(node, 2);
                forChildren(node, recurse);
            }
            return;
        }
        auto getIdx = [&](uint32_t s) {
            const double f = (centroid[(size_t)s * 3 + axis] - ce.mn[axis]) / dims[axis];
            const int idx = (int)std::floor(f * bins);
            return std::min(idx, bins - 1);
        };
        std::vector<size_t> count(bins, 0);
        std::vector<Box6> bbox(bins);
        for (uint32_t s : S) {
            const int idx = getIdx(s);
            count[idx]++;
            bbox[idx].merge(bb + (size_t)s * 6);
        }
        double min_cost = 1.7976931348623157e308;
        size_t split_bin = 0;
        const double node_area = node->bb.area();
        for (size_t i = 0; i + 1 < (size_t)bins; i++) {
            size_t a_count = 0, b_count = 0;
            Box6 a_bb, b_bb;
            for (size_t j = 0; j < i + 1; j++) {
                a_count += count[j];
                a_bb.merge(bbox[j]);
            }
            for (size_t j = i + 1; j < (size_t)bins; j++) {
                b_count += count[j];
                b_bb.merge(bbox[j]);
            }
            const double cost = 1.0 + ((double)a_count * a_bb.area() + (double)b_count * b_bb.area()) / node_area;
            if (cost < min_cost) {
                split_bin = i;
                min_cost = cost;
            }
        }
        if (min_cost > (double)S.size()) {
            if (S.size() > kMaxLeaf) {
                arbitrarySplit(node, 2);
                forChildren(node, recurse);
            }
            return;
        }
        std::unique_ptr<SahNode> A(new SahNode()), B(new SahNode());
        for (uint32_t s : S) {
            SahNode* side = (size_t)getIdx(s) <= split_bin ? A.get() : B.get();
            side->surfaces.push_back(s);
            side->bb.merge(bb + (size_t)s * 6);
        }
        std::vector<uint32_t>().swap(S);
        if (!A->surfaces.empty()) node->children.push_back(std::move(A));
        if (!B->surfaces.empty()) node->children.push_back(std::move(B));
        forChildren(node, recurse);
    }

    void quaternary(SahNode* node) {  // recursiveBuildQuaternarySAH, bvh.cpp:290-426
        auto& S = node->surfaces;
        if (S.size() <= kLeaf) return;
        Box6 ce;
        for (uint32_t s : S) ce.mergePoint(centroid + (size_t)s * 3);
        const double dims[3] = {ce.mx[0] - ce.mn[0], ce.mx[1] - ce.mn[1], ce.mx[2] - ce.mn[2]};
        int ax, ay;
        if (dims[0] > dims[1]) {
            ax = 0;
            ay = dims[1] > dims[2] ? 1 : 2;
        } else {
            if (dims[0] > dims[2]) { ax = 0; ay = 1; } else { ax = 1; ay = 2; }
        }
        if (dims[ax] < 1e-9 || dims[ay] < 1e-9) {  // one usable axis: the binary rule, same node (bvh.cpp:313-318)
            binary(node);
            return;
        }
        auto getIdx = [&](uint32_t s, int& ix, int& iy) {
            const double fx = (centroid[(size_t)s * 3 + ax] - ce.mn[ax]) / dims[ax];
            const double fy = (centroid[(size_t)s * 3 + ay] - ce.mn[ay]) / dims[ay];
            ix = std::min((int)std::floor(fx * (double)bins), bins - 1);
            iy = std::min((int)std::floor(fy * (double)bins), bins - 1);
        };
        std::vector<size_t> count((size_t)bins * bins, 0);
        std::vector<Box6> bbox((size_t)bins * bins);
        for (uint32_t s : S) {
            int ix, iy;
            getIdx(s, ix, iy);
            count[(size_t)ix * bins + iy]++;
            bbox[(size_t)ix * bins + iy].merge(bb + (size_t)s * 6);
        }
        double min_cost = 1.7976931348623157e308;
        int split_x = 0, split_y = 0;
        const double node_area = node->bb.area();
        for (int i = 0; i < bins - 1; i++)
            for (int j = 0; j < bins - 1; j++) {
                double cost = 0.0;
                for (int v = 0; v < 4; v++) {
                    const int x0 = (v & 1) ? i + 1 : 0, x1 = (v & 1) ? bins : i + 1;
                    const int y0 = (v & 2) ? j + 1 : 0, y1 = (v & 2) ? bins : j + 1;
                    size_t n = 0;
                    Box6 q;
                    for (int x = x0; x < x1; x++)
                        for (int y = y0; y < y1; y++) {
                            n += count[(size_t)x * bins + y];
                            q.merge(bbox[(size_t)x * bins + y]);
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
        auto recurse = [this](SahNode* n) { quaternary(n); };
        if (min_cost > (double)S.size()) {
            if (S.size() > kMaxLeaf) {
                arbitrarySplit(node, 4);
                forChildren(node, recurse);
            }
            return;
        }
        std::unique_ptr<SahNode> quad[4];
        for (uint32_t s : S) {
            int ix, iy;
            getIdx(s, ix, iy);
            const int ci = (ix > split_x ? 1 : 0) | (iy > split_y ? 2 : 0);
            if (!quad[ci]) quad[ci].reset(new SahNode());
            quad[ci]->surfaces.push_back(s);
            quad[ci]->bb.merge(bb + (size_t)s * 6);
        }
        std::vector<uint32_t>().swap(S);
        for (auto& q : quad)
            if (q) node->children.push_back(std::move(q));
        forChildren(node, recurse);
    }
};

// BVH::compact, bvh.cpp:428-449: depth-first numbering, surfaces of the leaves in visiting order.
size_t countNodes(const SahNode* n) {
    size_t c = 1;
    for (const auto& ch : n->children) c += countNodes(ch.get());
    return c;
}

void compact(const SahNode* node, uint32_t next_sibling, mcrt_bvh* B) {
    B->start.push_back((uint32_t)B->order.size());
    B->count.push_back((uint32_t)(uint8_t)node->surfaces.size());  // LinearNode::num_surfaces is a uint8_t (bvh.cpp:433)
    B->next.push_back(next_sibling);
    for (int c = 0; c < 3; c++) B->bounds.push_back(node->bb.mn[c]);
    for (int c = 0; c < 3; c++) B->bounds.push_back(node->bb.mx[c]);
    for (uint32_t s : node->surfaces) B->order.push_back(s);
    const size_t nc = node->children.size();
    for (size_t i = 0; i < nc; i++) {
        // a child's next sibling sits right after the child's subtree in depth-first order; 0 = none (bvh.cpp:443-447)
        const uint32_t my_id = (uint32_t)B->start.size();
        const uint32_t sibling = i + 1 < nc ? my_id + (uint32_t)countNodes(node->children[i].get()) : 0u;
        compact(node->children[i].get(), sibling, B);
    }
}

int buildSah(const mcrt_scene_desc* s, int arity, uint32_t bins, uint32_t threads, mcrt_bvh* B) {
    const uint64_t n = s->num_surfaces;
    std::vector<double> bb(n * 6), centroid(n * 3);
    for (uint64_t i = 0; i < n; i++) {
        double* b = &bb[i * 6];
        surfaceBounds(s->surf_kind[i], s->surf_v + 9 * i, s->quadrics, b);
        for (int c = 0; c < 3; c++) centroid[i * 3 + c] = (b[3 + c] + b[c]) / 2.0;  // BB().centroid()
    }
    SahBuilder sb;
    sb.bb = bb.data();
    sb.centroid = centroid.data();
    sb.bins = (int)(bins ? bins : (arity == 4 ? 8u : 16u));  // bvh.cpp:29,36
    unsigned hw = threads ? threads : std::thread::hardware_concurrency();
    sb.free_threads.store((int)(hw > 1 ? hw - 1 : 0));
    SahNode root;
    for (int c = 0; c < 3; c++) {  // root->BB = the scene box handed to the builder (bvh.cpp:20)
        root.bb.mn[c] = s->bb_min[c];
        root.bb.mx[c] = s->bb_max[c];
    }
    root.surfaces.resize(n);
    std::iota(root.surfaces.begin(), root.surfaces.end(), 0u);
    if (arity == 4) sb.quaternary(&root);
    else sb.binary(&root);
    compact(&root, 0, B);
    finishBvhDesc(B);
    return MCRT_OK;
}

}  // namespace

extern "C" {

int mcrt_bvh_build_octree(mcrt_ctx* ctx, const mcrt_scene_desc* scene, mcrt_bvh** out) {
    if (!out || !sceneUsable(scene)) return ctx ? ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_bvh_build_octree: bad scene descriptor") : MCRT_ERR_INVALID;
    mcrt_bvh* B = new mcrt_bvh();
    const int rc = ctx ? bvhOctreeGpu(ctx, scene, B) : buildHost(scene, B);
    if (rc != MCRT_OK) {
        delete B;
        if (ctx && rc == MCRT_ERR_UNSUPPORTED) return ctxFail(ctx, rc, "more than 8 surface centroids inside one 2^-21 cell of the scene cube");
        return rc;
    }
    *out = B;
    return MCRT_OK;
}

int mcrt_bvh_build_sah(const mcrt_scene_desc* scene, int arity, uint32_t bins_per_axis, uint32_t threads, mcrt_bvh** out) {
    if (!out || !sceneUsable(scene) || (arity != 2 && arity != 4) || bins_per_axis == 1 || bins_per_axis > 1024) return MCRT_ERR_INVALID;
    mcrt_bvh* B = new mcrt_bvh();
    const int rc = buildSah(scene, arity, bins_per_axis, threads, B);
    if (rc != MCRT_OK) {
        delete B;
        return rc;
    }
    *out = B;
    return MCRT_OK;
}

int mcrt_bvh_build_sah_gpu(mcrt_ctx* ctx, const mcrt_scene_desc* scene, int arity, uint32_t bins_per_axis, mcrt_bvh** out) {
    const uint32_t bins = bins_per_axis ? bins_per_axis : (arity == 4 ? 8u : 16u);  // bvh.cpp:29,36
    if (!out || !sceneUsable(scene) || (arity != 2 && arity != 4) || bins < 2)
        return ctx ? ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_bvh_build_sah_gpu: bad scene descriptor, arity or bin count") : MCRT_ERR_INVALID;
    if (bins > kSahMaxBins) {
        // The level loop keeps a table of bins (binary) or bins x bins (quaternary) counters and boxes per open node, sized for the
        // reference's defaults (16 and 8 x 8, bvh.cpp:29,36) and up to 16. The reference takes any "bins_per_axis" (bvh.cpp:24-40): more
        // than 16 goes to the recursive builder on all host threads, which has no limit and builds the same tree (same split rule, same
        // order) - the caller gets its hierarchy either way (until round 4: MCRT_ERR_UNSUPPORTED).
        if (bins > 1024) return ctx ? ctxFail(ctx, MCRT_ERR_INVALID, "mcrt_bvh_build_sah_gpu: more than 1024 bins per axis") : MCRT_ERR_INVALID;
        const int rc = mcrt_bvh_build_sah(scene, arity, bins, 0, out);
        return (rc != MCRT_OK && ctx) ? ctxFail(ctx, rc, "mcrt_bvh_build_sah_gpu: host build with more than 16 bins per axis failed") : rc;
    }
    mcrt_bvh* B = nullptr;
    try {  // (no exception may cross the C boundary: an allocation failure in the level loop is an error code, and B is freed)
        B = new mcrt_bvh();
        int rc;
        if (ctx) {
            rc = bvhSahGpu(ctx, scene, arity, (int)bins, B);
        } else {  // the same level-synchronous build with its passes as host loops
            const uint64_t n = scene->num_surfaces;
            std::vector<double> bb(n * 6), centroid(n * 3);
            for (uint64_t i = 0; i < n; i++) {
                surfaceBounds(scene->surf_kind[i], scene->surf_v + 9 * i, scene->quadrics, &bb[i * 6]);
                for (int c = 0; c < 3; c++) centroid[i * 3 + c] = (bb[i * 6 + 3 + c] + bb[i * 6 + c]) / 2.0;  // BB().centroid()
            }
            rc = buildSahLevelsHost(bb.data(), centroid.data(), n, scene->bb_min, scene->bb_max, arity, (int)bins, B);
        }
        if (rc != MCRT_OK) {
            delete B;
            return rc;
        }
    } catch (...) {
        delete B;
        return ctx ? ctxFail(ctx, MCRT_ERR_HIP, "mcrt_bvh_build_sah_gpu: out of memory") : MCRT_ERR_HIP;
    }
    *out = B;
    return MCRT_OK;
}

const mcrt_bvh_desc* mcrt_bvh_get(const mcrt_bvh* bvh) { return bvh ? &bvh->desc : nullptr; }
void mcrt_bvh_free(mcrt_bvh* bvh) { delete bvh; }

int mcrt_scene_with_bvh(const mcrt_scene_desc* s, const mcrt_bvh_desc* bvh, mcrt_scene** out) {
    if (!out) return MCRT_ERR_INVALID;
    *out = nullptr;
    if (!s || !bvh || bvh->num_surfaces != s->num_surfaces || !bvh->order || !s->surf_kind || !s->surf_v || !s->surf_interpolate ||
        !s->surf_material || !s->surf_area || !s->surf_e || !s->materials || s->num_materials == 0)
        return MCRT_ERR_INVALID;
    if (bvh->num_nodes && (!bvh->node_bounds || !bvh->node_start_surface || !bvh->node_num_surfaces || !bvh->node_next_sibling)) return MCRT_ERR_INVALID;
    if (s->num_lights && (!s->light_surface || !s->light_cdf)) return MCRT_ERR_INVALID;
    if (s->num_quadrics && !s->quadrics) return MCRT_ERR_INVALID;
    const size_t n = s->num_surfaces;
    for (uint32_t i = 0; i < s->num_lights; i++)
        if (s->light_surface[i] >= n) return MCRT_ERR_INVALID;
    mcrt_scene* S = nullptr;
    try {
        S = new mcrt_scene();
        S->kind.resize(n);
        S->interp.resize(n);
        S->material.resize(n);
        S->area.resize(n);
        S->v.resize(n * 9);
        S->e.resize(n * 9);
        if (s->surf_vn) S->vn.resize(n * 9);
        std::vector<uint32_t> where(n, 0xFFFFFFFFu);  // input index -> new position
        for (size_t i = 0; i < n; i++) {
            const size_t src = bvh->order[i];
            if (src >= n || where[src] != 0xFFFFFFFFu) {  // order must be a permutation
                delete S;
                return MCRT_ERR_INVALID;
            }
            where[src] = (uint32_t)i;
            S->kind[i] = s->surf_kind[src];
            S->interp[i] = s->surf_interpolate[src];
            S->material[i] = s->surf_material[src];
            S->area[i] = s->surf_area[src];
            memcpy(&S->v[i * 9], s->surf_v + src * 9, 72);
            memcpy(&S->e[i * 9], s->surf_e + src * 9, 72);
            if (s->surf_vn) memcpy(&S->vn[i * 9], s->surf_vn + src * 9, 72);
        }
        S->light_surface.resize(s->num_lights);
        for (uint32_t i = 0; i < s->num_lights; i++) S->light_surface[i] = where[s->light_surface[i]];
        // the result owns everything: the caller may free `bvh` and the source descriptor's arrays afterwards
        const size_t nn = bvh->num_nodes;
        S->node_bounds.assign(bvh->node_bounds, bvh->node_bounds + nn * 6);
        S->node_start.assign(bvh->node_start_surface, bvh->node_start_surface + nn);
        S->node_count.assign(bvh->node_num_surfaces, bvh->node_num_surfaces + nn);
        S->node_next.assign(bvh->node_next_sibling, bvh->node_next_sibling + nn);
        S->materials.assign(s->materials, s->materials + s->num_materials);
        if (s->num_lights) S->light_cdf.assign(s->light_cdf, s->light_cdf + s->num_lights);
        if (s->num_quadrics) S->quadrics.assign(s->quadrics, s->quadrics + (size_t)s->num_quadrics * 22);
    } catch (...) {
        delete S;
        return MCRT_ERR_INVALID;
    }
    S->desc = *s;  // scalars and counts
    S->desc.num_nodes = bvh->num_nodes;
    S->desc.node_bounds = S->node_bounds.data();
    S->desc.node_start_surface = S->node_start.data();
    S->desc.node_num_surfaces = S->node_count.data();
    S->desc.node_next_sibling = S->node_next.data();
    S->desc.surf_kind = S->kind.data();
    S->desc.surf_interpolate = S->interp.data();
    S->desc.surf_material = S->material.data();
    S->desc.surf_area = S->area.data();
    S->desc.surf_v = S->v.data();
    S->desc.surf_e = S->e.data();
    S->desc.surf_vn = s->surf_vn ? S->vn.data() : nullptr;
    S->desc.materials = S->materials.data();
    S->desc.light_surface = S->light_surface.data();
    S->desc.light_cdf = s->num_lights ? S->light_cdf.data() : nullptr;
    S->desc.quadrics = s->num_quadrics ? S->quadrics.data() : nullptr;
    *out = S;
    return MCRT_OK;
}

const mcrt_scene_desc* mcrt_scene_get(const mcrt_scene* scene) { return scene ? &scene->desc : nullptr; }
void mcrt_scene_free(mcrt_scene* scene) { delete scene; }

}  // extern "C"
