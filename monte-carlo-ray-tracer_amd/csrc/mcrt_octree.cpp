// Host-side photon-map builder: photon list -> the linear octree LinearOctree<Photon> describes
// (include/mcrt.h: mcrt_photon_map_desc). The reference builds a pointer octree by serial insertion
// (octree/octree.cpp:35-80, leaf capacity max_node_data, octant = per-axis `pos >= centroid`) and
// compacts it depth-first with tight bounding boxes, skipping empty octants
// (octree/linear-octree.cpp:202-244). A node ends up a leaf iff at most max_node_data photons fell
// into its cell, so the same tree is obtained top-down by partitioning; only the order of photons
// inside a leaf (insertion order in the reference, input order here) differs, which no query result
// depends on. Builders are host code in the reference and stay host code here.
#include <cstdint>
#include <cstring>
#include <vector>

#include "mcrt_octree_shared.hpp"

namespace {

struct Cell {
    double mn[3], mx[3];
};

struct Builder {
    const float* in;
    uint32_t max_node_data;
    mcrt_photon_map* M;

    // returns the number of photons in the subtree; `idx` are indices into `in`
    uint64_t compact(std::vector<uint64_t>& idx, const Cell& cell, bool last, int depth, double bb_out[6]) {
        const uint32_t node = (uint32_t)M->start.size();
        M->start.push_back(M->photons.size() / 8);
        M->contained.push_back(0);
        M->next.push_back(0);
        M->leaf.push_back(0);
        M->bounds.resize(M->bounds.size() + 6);
        double bb[6] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
                        -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
        const bool is_leaf = idx.size() <= max_node_data || depth > mcrt::kMaxOctreeDepth;
        uint64_t contained = 0;
        if (is_leaf) {
            for (uint64_t i : idx) {
                const float* p = in + i * 8;
                for (int c = 0; c < 3; c++) {  // BoundingBox::merge(pos), bounding-box.cpp:66-73
                    const double v = (double)p[3 + c];
                    if (bb[c] > v) bb[c] = v;
                    if (bb[3 + c] < v) bb[3 + c] = v;
                }
                M->photons.insert(M->photons.end(), p, p + 8);
            }
            contained = idx.size();
            std::vector<uint64_t>().swap(idx);
        } else {
            // split, octree.cpp:46-58: centroid = (max + min) / 2, half = dimensions / 2
            double origin[3], half[3];
            for (int c = 0; c < 3; c++) {
                origin[c] = (cell.mx[c] + cell.mn[c]) / 2.0;
                half[c] = (cell.mx[c] - cell.mn[c]) / 2.0;
            }
            std::vector<uint64_t> part[8];
            for (uint64_t i : idx) {  // insertInOctant, octree.cpp:71-80
                const float* p = in + i * 8;
                int o = 0;
                for (int c = 0; c < 3; c++)
                    if ((double)p[3 + c] >= origin[c]) o |= (4 >> c);
                part[o].push_back(i);
            }
            std::vector<uint64_t>().swap(idx);
            int last_used = -1;
            for (int o = 0; o < 8; o++)
                if (!part[o].empty()) last_used = o;
            for (int o = 0; o < 8; o++) {
                if (part[o].empty()) continue;  // linear-octree.cpp:225-231: empty leaf octants are dropped
                Cell child;
                for (int c = 0; c < 3; c++) {
                    double no = origin[c] + half[c] * ((o & (4 >> c)) ? 0.5 : -0.5);
                    child.mn[c] = no - half[c] * 0.5;
                    child.mx[c] = no + half[c] * 0.5;
                }
                double cbb[6];
                contained += compact(part[o], child, o == last_used, depth + 1, cbb);
                for (int c = 0; c < 3; c++) {  // BoundingBox::merge(BB), bounding-box.cpp:57-64
                    if (bb[c] > cbb[c]) bb[c] = cbb[c];
                    if (bb[3 + c] < cbb[3 + c]) bb[3 + c] = cbb[3 + c];
                }
            }
        }
        M->leaf[node] = is_leaf ? 1 : 0;
        M->contained[node] = contained;
        M->next[node] = last ? 0xFFFFFFFFu : (uint32_t)M->start.size();
        memcpy(&M->bounds[(size_t)node * 6], bb, sizeof(bb));
        memcpy(bb_out, bb, sizeof(bb));
        return contained;
    }
};

}  // namespace

extern "C" {

int mcrt_photon_map_build(const float* photons, uint64_t num_photons, const double bb_min[3], const double bb_max[3],
                          uint32_t max_photons_per_leaf, mcrt_photon_map** out) {
    if (!out || (num_photons && !photons) || !bb_min || !bb_max || max_photons_per_leaf == 0) return MCRT_ERR_INVALID;
    mcrt_photon_map* M = new mcrt_photon_map();
    if (num_photons) {
        M->photons.reserve((size_t)num_photons * 8);
        std::vector<uint64_t> idx(num_photons);
        for (uint64_t i = 0; i < num_photons; i++) idx[i] = i;
        Cell root;
        for (int c = 0; c < 3; c++) {
            root.mn[c] = bb_min[c];
            root.mx[c] = bb_max[c];
        }
        Builder b{photons, max_photons_per_leaf, M};
        double bb[6];
        b.compact(idx, root, true, 0, bb);
    }
    mcrt::finishMapDesc(M);
    *out = M;
    return MCRT_OK;
}

const mcrt_photon_map_desc* mcrt_photon_map_get(const mcrt_photon_map* map) { return map ? &map->desc : nullptr; }

void mcrt_photon_map_free(mcrt_photon_map* map) { delete map; }

}  // extern "C"
