// Record lists of the wave-cooperative photon search (WideRec, mcrt_waveknn.hpp) for a map that arrives as the reference's linear
// octree (octants in depth-first order: the children of octant o are o + 1 and its next_sibling chain, linear-octree.cpp:119-170).
// Plain host C++: used by mcrt_upload_photons (mcrt_hip.hip) and by the host emulation of the wave search (tests/emu/wave_knn_emu.cpp);
// the device-built maps make the same lists with kernels (mcrt_photon_device.hpp).
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mcrt.h"
#include "mcrt_waveknn.hpp"

namespace mcrt {

// For every inner octant (not a leaf, more than k photons) the list of what one step of the descent tests: its children that can be
// scanned right away (leaves, or <= k photons: linear-octree.cpp:51) and, for every other child, that child's children. `contained` =
// the octants' photon counts as uint32. Returns 0, or 1 (an octant with more than 8 children: not an octree) / 2 (more than 2^32
// records); root_a / root_m = the root as an entry.
inline int buildWideRecords(const mcrt_photon_map_desc* m, const uint32_t* contained, uint32_t k, std::vector<WideRec>& wide, uint32_t& root_a,
                            uint32_t& root_m) {
    const size_t n = m->num_octants;
    auto scannable = [&](size_t o) { return m->octant_leaf[o] != 0 || contained[o] <= k; };
    auto forChildren = [&](size_t o, auto f) {
        uint32_t c = (uint32_t)o + 1;
        int count = 0;
        while (c != 0xFFFFFFFFu && c < n) {
            if (++count > 8) return false;
            f((size_t)c);
            c = m->octant_next_sibling[c];
        }
        return true;
    };
    std::vector<uint32_t> first(n, 0), count(n, 0);
    uint64_t total = 0;
    bool ok = true;
    for (size_t o = 0; o < n && ok; o++) {
        if (scannable(o)) continue;
        uint32_t cnt = 0;
        ok = forChildren(o, [&](size_t c) {
            if (scannable(c)) cnt++;
            else ok = forChildren(c, [&](size_t) { cnt++; }) && ok;
        }) && ok;
        first[o] = (uint32_t)total;
        count[o] = cnt;
        total += cnt;
    }
    if (!ok) return 1;
    if (total > 0xFFFFFFFFull) return 2;
    wide.assign(total, WideRec{});
    auto fill = [&](WideRec& r, size_t c) {
        memset(&r, 0, sizeof(r));
        memcpy(r.b, m->octant_bounds + c * 6, 48);
        r.contained = contained[c];
        if (scannable(c)) {
            r.a = (uint32_t)m->octant_start_data[c];
            r.m = 0x80000000u | contained[c];
        } else {
            r.a = first[c];
            r.m = count[c];
        }
    };
    for (size_t o = 0; o < n; o++) {
        if (scannable(o)) continue;
        size_t at = first[o];
        forChildren(o, [&](size_t c) {
            if (scannable(c)) fill(wide[at++], c);
            else forChildren(c, [&](size_t g) { fill(wide[at++], g); });
        });
    }
    root_a = scannable(0) ? 0u : first[0];
    root_m = scannable(0) ? (0x80000000u | contained[0]) : count[0];
    return 0;
}

}  // namespace mcrt
