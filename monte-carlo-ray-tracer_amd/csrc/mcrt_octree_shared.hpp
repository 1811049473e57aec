// Pieces shared by the host photon-map builder (mcrt_octree.cpp), the GPU-assisted builder
// (mcrt_octree_gpu.hip) and the CPU test harness (tests/emu): the owning map object, the per-photon cell
// code and the assembly of the linear octree from sorted codes.
//
// The reference inserts photons one at a time into a pointer octree (octree/octree.cpp:35-80: a leaf that
// exceeds max_node_data is split at the centre of its cell, octant bit per axis = `pos >= centre`) and
// compacts it depth-first, dropping empty octants, with tight boxes (octree/linear-octree.cpp:202-244).
// Which leaf a photon ends up in depends only on its position: at every level the octant is decided by
// comparing with the cell centre, and the child cell follows from the parent cell by the same FP64
// expressions. So a photon's whole root-to-depth-21 path can be computed independently of all others
// (photonCellCode, 3 bits per level, most significant first), sorting by that code puts the photons in the
// depth-first order of the compacted tree (subtree data contiguous, octants in index order), and the octants
// are the code prefixes that hold more than max_node_data photons (inner) or their non-empty children.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_math.hpp"

struct mcrt_photon_map {
    std::vector<double> bounds;
    std::vector<uint64_t> start, contained;
    std::vector<uint32_t> next;
    std::vector<uint8_t> leaf;
    std::vector<float> photons;
    mcrt_photon_map_desc desc;
};

namespace mcrt {

constexpr int kCodeLevels = 21;   // 63-bit codes
constexpr int kMaxOctreeDepth = 60;  // the host builder's recursion guard (same rule here)

// Root-to-level-21 octant path of a position, cells exactly as octree.cpp:46-58,71-80.
MCRT_HD unsigned long long cellCode(double px, double py, double pz, const double* bb_min, const double* bb_max) {
    const double pos[3] = {px, py, pz};
    double mn[3] = {bb_min[0], bb_min[1], bb_min[2]}, mx[3] = {bb_max[0], bb_max[1], bb_max[2]};
    unsigned long long code = 0ull;
    for (int level = 0; level < kCodeLevels; level++) {
        unsigned o = 0;
        for (int c = 0; c < 3; c++) {
            const double origin = (mx[c] + mn[c]) / 2.0, half = (mx[c] - mn[c]) / 2.0;
            const bool up = pos[c] >= origin;
            if (up) o |= (4u >> c);
            const double no = origin + half * (up ? 0.5 : -0.5);
            mn[c] = no - half * 0.5;
            mx[c] = no + half * 0.5;
        }
        code = (code << 3) | o;
    }
    return code;
}
MCRT_HD unsigned long long photonCellCode(const float* ph, const double* bb_min, const double* bb_max) {  // position = ph[3..5]
    return cellCode((double)ph[3], (double)ph[4], (double)ph[5], bb_min, bb_max);
}

inline void finishMapDesc(mcrt_photon_map* M) {
    memset(&M->desc, 0, sizeof(M->desc));
    M->desc.num_octants = (uint32_t)M->start.size();
    M->desc.octant_bounds = M->bounds.data();
    M->desc.octant_start_data = M->start.data();
    M->desc.octant_contained_data = M->contained.data();
    M->desc.octant_next_sibling = M->next.data();
    M->desc.octant_leaf = M->leaf.data();
    M->desc.num_photons = M->photons.size() / 8;
    M->desc.photons = M->photons.data();
}

// Octants (depth-first, start/contained/next_sibling/leaf; boxes left empty) from the sorted codes.
struct OctreeAssembler {
    const unsigned long long* keys;
    uint32_t max_node_data;
    mcrt_photon_map* M;
    std::vector<uint32_t> parent, leaves;  // parent octant of every octant; indices of the leaf octants
    bool too_deep = false;                 // a level-21 cell still holds more than max_node_data photons

    void node(uint64_t lo, uint64_t hi, int depth, bool last, uint32_t par) {
        const uint32_t id = (uint32_t)M->start.size();
        M->start.push_back(lo);
        M->contained.push_back(hi - lo);
        M->next.push_back(0);
        M->leaf.push_back(0);
        parent.push_back(par);
        const bool is_leaf = (hi - lo) <= max_node_data || depth > kMaxOctreeDepth;
        if (is_leaf || depth >= kCodeLevels) {
            if (!is_leaf) too_deep = true;
            M->leaf[id] = 1;
            leaves.push_back(id);
        } else {
            const int shift = 3 * (kCodeLevels - 1 - depth);
            uint64_t cut[9];
            cut[0] = lo;
            for (unsigned o = 0; o < 8; o++) {  // first key of [cut[o], hi) whose octant at this level exceeds o
                uint64_t a = cut[o], b = hi;
                while (a < b) {
                    const uint64_t mid = a + (b - a) / 2;
                    if (((keys[mid] >> shift) & 7ull) <= o) a = mid + 1;
                    else b = mid;
                }
                cut[o + 1] = a;
            }
            int last_used = -1;
            for (int o = 0; o < 8; o++)
                if (cut[o + 1] > cut[o]) last_used = o;
            for (int o = 0; o < 8; o++)
                if (cut[o + 1] > cut[o]) node(cut[o], cut[o + 1], depth + 1, o == last_used, id);  // empty octants are dropped
        }
        M->next[id] = last ? 0xFFFFFFFFu : (uint32_t)M->start.size();
    }

    // leaf boxes are in M->bounds already: merge them upwards (BoundingBox::merge, bounding-box.cpp:57-64)
    void mergeBounds() {
        const uint32_t n = (uint32_t)M->start.size();
        for (uint32_t i = 0; i < n; i++)
            if (!M->leaf[i])
                for (int c = 0; c < 3; c++) {
                    M->bounds[(size_t)i * 6 + c] = 1.7976931348623157e308;
                    M->bounds[(size_t)i * 6 + 3 + c] = -1.7976931348623157e308;
                }
        for (uint32_t i = n; i-- > 1;) {
            double* p = &M->bounds[(size_t)parent[i] * 6];
            const double* b = &M->bounds[(size_t)i * 6];
            for (int c = 0; c < 3; c++) {
                if (p[c] > b[c]) p[c] = b[c];
                if (p[3 + c] < b[3 + c]) p[3 + c] = b[3 + c];
            }
        }
    }
};

}  // namespace mcrt
