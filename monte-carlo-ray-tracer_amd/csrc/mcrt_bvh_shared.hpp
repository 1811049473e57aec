// Octree BVH builder pieces shared by the host path (mcrt_bvh.cpp), the GPU-assisted path (mcrt_octree_gpu.hip) and
// the CPU test harness.
//
// The reference's default BVH ("type": "octree", bvh/bvh.cpp:41-56,130-163) is an Octree<SurfaceCentroid> over the
// cube around Scene::BB() with leaves of at most 8 surfaces, filled by serial insertion of the surfaces' box centroids,
// then converted: one BVH node per non-empty octant, box = union of the surfaces' boxes below it, leaves list their
// surfaces in insertion order, nodes and surfaces numbered depth-first (BVH::compact, bvh.cpp:428-449). It is the same
// Octree class as the photon map's, so the same observation holds (mcrt_octree_shared.hpp): a surface's leaf depends
// only on its centroid, the root-to-depth-21 octant path of every centroid can be computed independently, sorting by
// it gives the depth-first order, and the nodes are the path prefixes that hold more than 8 surfaces (inner) or their
// non-empty children. Re-sorting the surfaces of each leaf by input index restores insertion order, so the result is
// the reference's tree bit for bit: same nodes, boxes, links and surface order.
#pragma once

#include <algorithm>

#include "mcrt_octree_shared.hpp"
#include "mcrt_scene.hpp"

struct mcrt_bvh {
    std::vector<double> bounds;
    std::vector<uint32_t> start, count, next, order;
    mcrt_bvh_desc desc;
};

namespace mcrt {

constexpr uint32_t kBvhLeafSurfaces = 8;  // BVH::leaf_surfaces, bvh/bvh.hpp:91

// Surface::Base::BB(): triangle.cpp:115-122 (merge of the vertices), sphere.cpp:56-62, quadric.cpp:36,117-118 (BB_)
MCRT_HD void surfaceBounds(uint8_t kind, const double* v /* surf_v + 9 i */, const double* quadrics, double* bb) {
    if (kind == MCRT_SURF_SPHERE) {
        for (int c = 0; c < 3; c++) {
            bb[c] = v[c] - v[3];
            bb[3 + c] = v[c] + v[3];
        }
    } else if (kind == MCRT_SURF_QUADRIC) {
        const double* q = quadrics + (size_t)v[0] * 22;
        for (int c = 0; c < 6; c++) bb[c] = q[16 + c];
    } else {
        for (int c = 0; c < 3; c++) {
            bb[c] = 1.7976931348623157e308;
            bb[3 + c] = -1.7976931348623157e308;
        }
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < 3; c++) {
                const double x = v[3 * k + c];
                if (bb[c] > x) bb[c] = x;
                if (bb[3 + c] < x) bb[3 + c] = x;
            }
    }
}

// The octree's root cell: the cube around the scene box (bvh.cpp:45-46).
inline void bvhRootCube(const mcrt_scene_desc* s, double* mn, double* mx) {
    double dims[3], half_max;
    for (int c = 0; c < 3; c++) dims[c] = s->bb_max[c] - s->bb_min[c];
    half_max = gmax(gmax(dims[0], dims[1]), dims[2]) / 2.0;
    for (int c = 0; c < 3; c++) {
        const double centroid = (s->bb_max[c] + s->bb_min[c]) / 2.0;
        mn[c] = centroid - half_max;
        mx[c] = centroid + half_max;
    }
}

inline void finishBvhDesc(mcrt_bvh* B) {
    memset(&B->desc, 0, sizeof(B->desc));
    B->desc.num_nodes = (uint32_t)B->start.size();
    B->desc.node_bounds = B->bounds.data();
    B->desc.node_start_surface = B->start.data();
    B->desc.node_num_surfaces = B->count.data();
    B->desc.node_next_sibling = B->next.data();
    B->desc.num_surfaces = (uint32_t)B->order.size();
    B->desc.order = B->order.data();
}

// From the surfaces' sorted cell codes (`keys`, with `index[i]` = input index of the i-th sorted surface, equal codes in
// input order) and their boxes in sorted order: the reference's LinearNode arrays and the surface order.
// Returns false when more than 8 centroids share one 2^-21 cell (the reference would recurse further).
inline bool assembleOctreeBvh(const unsigned long long* keys, const uint32_t* index, const double* sorted_bb, uint64_t n, mcrt_bvh* B) {
    mcrt_photon_map M;  // the assembler's octant arrays
    OctreeAssembler A;
    A.keys = keys;
    A.max_node_data = kBvhLeafSurfaces;
    A.M = &M;
    A.node(0, n, 0, true, 0xFFFFFFFFu);
    if (A.too_deep) return false;
    const uint32_t nodes = (uint32_t)M.start.size();
    B->start.resize(nodes);
    B->count.resize(nodes);
    B->next.resize(nodes);
    for (uint32_t i = 0; i < nodes; i++) {
        B->start[i] = (uint32_t)M.start[i];
        B->count[i] = M.leaf[i] ? (uint32_t)M.contained[i] : 0u;          // LinearNode::num_surfaces: leaves only
        B->next[i] = M.next[i] == 0xFFFFFFFFu ? 0u : M.next[i];          // 0 = no sibling (bvh.cpp:447)
    }
    B->order.assign(index, index + n);
    M.bounds.assign((size_t)nodes * 6, 0.0);
    for (uint32_t l : A.leaves) {
        double* bb = &M.bounds[(size_t)l * 6];
        for (int c = 0; c < 3; c++) {
            bb[c] = 1.7976931348623157e308;
            bb[3 + c] = -1.7976931348623157e308;
        }
        const uint64_t lo = M.start[l], hi = lo + M.contained[l];
        for (uint64_t i = lo; i < hi; i++) {  // BoundingBox::merge(BB), bounding-box.cpp:56-63
            const double* sb = sorted_bb + i * 6;
            for (int c = 0; c < 3; c++) {
                if (bb[c] > sb[c]) bb[c] = sb[c];
                if (bb[3 + c] < sb[3 + c]) bb[3 + c] = sb[3 + c];
            }
        }
        std::sort(B->order.begin() + lo, B->order.begin() + hi);  // insertion order inside a leaf = input order
    }
    A.mergeBounds();
    B->bounds = M.bounds;
    finishBvhDesc(B);
    return true;
}

}  // namespace mcrt
