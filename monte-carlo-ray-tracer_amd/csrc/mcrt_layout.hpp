// Host-side conversion of the reference's flattened scene (include/mcrt.h descriptors) into the
// layout the gfx950 kernels read (mcrt_scene.hpp). Used by mcrt_upload_scene; host only.
#pragma once

#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_scene.hpp"

namespace mcrt {

struct HostLayout {
    std::vector<double> node_bounds;   // [n][6], breadth-first, children contiguous
    std::vector<NodeMeta> node_meta;   // [n]
    std::vector<double> prim;          // [n][kPrimStride]
    std::vector<double> normal;        // [n][3] Triangle::normal_
    bool any_vn = false;
    // kind-sorted copy for the flat (tiny-scene) loop: triangles first, then spheres
    std::vector<double> flat_prim;     // [n][kPrimStride]
    std::vector<uint32_t> flat_index;  // [n] sorted slot -> surface index
    uint32_t flat_tris = 0;
    std::vector<Node64> nodes64;       // [n] box + meta records, same order as node_bounds
};

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

inline int buildLayout(const mcrt_scene_desc* s, HostLayout& L, std::string& err) {
    const size_t ns = s->num_surfaces;
    L.prim.assign(ns * kPrimStride, 0.0);
    L.normal.assign(ns * 3, 0.0);
    L.any_vn = false;
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
        } else {
            err = "unsupported surface kind (only triangles and spheres)";
            return MCRT_ERR_UNSUPPORTED;
        }
    }
    for (uint32_t i = 0; i < s->num_lights; i++)
        if (s->light_surface[i] >= s->num_surfaces) {
            err = "light surface index out of range";
            return MCRT_ERR_INVALID;
        }
    L.flat_prim.assign(ns * kPrimStride, 0.0);
    L.flat_index.assign(ns, 0u);
    L.flat_tris = 0;
    size_t slot = 0;
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < ns; i++) {
            const bool sphere = s->surf_kind[i] == MCRT_SURF_SPHERE;
            if (sphere != (pass == 1)) continue;
            memcpy(&L.flat_prim[slot * kPrimStride], &L.prim[i * kPrimStride], kPrimStride * sizeof(double));
            L.flat_index[slot] = (uint32_t)i;
            slot++;
            if (!sphere) L.flat_tris++;
        }
    if (int rc = convertNodes(s, L.node_bounds, L.node_meta, err)) return rc;
    L.nodes64.assign(s->num_nodes, Node64{});
    for (uint32_t i = 0; i < s->num_nodes; i++) {
        Node64& n = L.nodes64[i];
        memcpy(n.b, &L.node_bounds[(size_t)i * 6], 48);
        const NodeMeta& m = L.node_meta[i];
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
    return MCRT_OK;
}

}  // namespace mcrt
