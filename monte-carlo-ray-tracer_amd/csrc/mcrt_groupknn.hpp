// mcrt_groupknn.hpp — LinearOctree::knnSearch (octree/linear-octree.cpp:25-117) for FOUR queries per wave, one per row
// of 16 lanes. Device code only.
//
// With one query per wave (mcrt_waveknn.hpp) a search is a chain of ~100 dependent steps — pop, one memory round trip,
// test, reduce, push — in which most instructions have something to do for a fraction of the lanes and the scalar unit
// carries the bookkeeping; the kernel sits at 57 % VALU busy with four waves per SIMD because there is nothing else to
// issue while a chain waits. Here the four DPP rows of a wave run four independent searches in lock step: every per-query
// scalar of the wave version (bound, candidate count, current octant, histogram scale) becomes a value that is uniform
// within a row, reductions stay inside a row (four DPP steps, no row broadcasts), and one instruction stream carries four
// chains. Same pruning rules, same candidates, same k-set as the wave version:
//   * a step visits the current octant of every row: rows at an inner octant test its record list (up to 64 records,
//     16 per pass), rows at a scannable octant scan its photons (16 per pass, four passes per round trip);
//   * the frontier is four entries per lane (64 per query); a record that finds its tester's slots taken goes to any free
//     slot of the row; a row whose frontier is full gives up and the caller repeats that query with waveKnnSearch;
//   * candidates go to a per-row buffer (96 entries) and a per-row histogram of distance2 (64 bins over the bound in force
//     at the first scan); the bound after a scan is the upper edge of the bin that holds the k-th nearest so far, the
//     buffer is reduced by a search for a coarse k-th key when it fills (all rows that need it at once), and the final
//     k-set is "coarse k-th key, then drop the largest until k are left" — entries equal to the k-th key stay in buffer
//     order until k are reached, as in waveSelectK and the reference's heap (linear-octree.cpp:58-77).
#ifndef MCRT_GROUPKNN_HPP
#define MCRT_GROUPKNN_HPP

#if defined(__HIPCC__) || defined(MCRT_WAVE_EMU)

#include "mcrt_waveknn.hpp"

namespace mcrt {

constexpr uint32_t kGrpLanes = 16, kGroups = 4;
constexpr uint32_t kGrpCand = 96;                      // candidate buffer entries per query
constexpr uint32_t kGrpRows = kGrpCand / kGrpLanes;    // ... = 6 per lane (entry j belongs to lane j % 16)
constexpr uint32_t kGrpHist = 64;                      // histogram bins per query (4 per lane)
constexpr uint32_t kGrpFront = 4;                      // frontier entries per lane
constexpr uint32_t kGrpMaxK = 64;                      // k + a pass of 16 candidates + slack must fit the buffer
constexpr uint32_t kGrpBytes = kGrpCand * 12u + kGrpHist * 4u + 16u;
constexpr uint32_t kGroupKnnBytes = kGroups * kGrpBytes;  // LDS per wave (>= kWaveKnnBytes: the wave version's buffers overlay it)
static_assert(kGroupKnnBytes >= kWaveKnnBytes, "the wave search must fit the same LDS");

struct GroupKnnLds {  // the regions of THIS lane's row
    MCRT_LDS_AS double* d2;
    MCRT_LDS_AS uint32_t* idx;
    MCRT_LDS_AS uint32_t* hist;
    MCRT_LDS_AS double* mail;  // one value handed from a lane to its row
};
__device__ inline GroupKnnLds groupKnnLds(MCRT_LDS_AS unsigned char* wave_base, uint32_t group) {
    MCRT_LDS_AS unsigned char* b = wave_base + group * kGrpBytes;
    GroupKnnLds G;
    G.d2 = reinterpret_cast<MCRT_LDS_AS double*>(b);
    G.idx = reinterpret_cast<MCRT_LDS_AS uint32_t*>(b + kGrpCand * 8u);
    G.hist = reinterpret_cast<MCRT_LDS_AS uint32_t*>(b + kGrpCand * 12u);
    G.mail = reinterpret_cast<MCRT_LDS_AS double*>(b + kGrpCand * 12u + kGrpHist * 4u);
    return G;
}
__device__ inline WaveKnnLds waveKnnLdsOver(MCRT_LDS_AS unsigned char* wave_base) {  // the wave version's buffers in the same bytes
    WaveKnnLds W;
    W.d2 = reinterpret_cast<MCRT_LDS_AS double*>(wave_base);
    W.idx = reinterpret_cast<MCRT_LDS_AS uint32_t*>(wave_base + kWaveCand * 8u);
    W.hist = reinterpret_cast<MCRT_LDS_AS uint32_t*>(wave_base + kWaveCand * 12u);
    return W;
}

// ---- reductions over a row of 16 lanes: butterflies through the DPP network, every lane of the row gets the result.
// All 64 lanes must execute them (rows that have nothing to reduce pass neutral values).
#define MCRT_ROW_REDUCE(v, OP)                                                                              \
    do {                                                                                                    \
        uint32_t t_;                                                                                        \
        t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0xB1, 0xF, 0xF, false); /* quad_perm [1,0,3,2] */ \
        (v) = OP((v), t_);                                                                                  \
        t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x4E, 0xF, 0xF, false); /* quad_perm [2,3,0,1] */ \
        (v) = OP((v), t_);                                                                                  \
        t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x141, 0xF, 0xF, false); /* row_half_mirror */    \
        (v) = OP((v), t_);                                                                                  \
        t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x140, 0xF, 0xF, false); /* row_mirror */         \
        (v) = OP((v), t_);                                                                                  \
    } while (0)
#define MCRT_OP_MIN(a, b) ((b) < (a) ? (b) : (a))
#define MCRT_OP_MAX(a, b) ((b) > (a) ? (b) : (a))
#define MCRT_OP_ADD(a, b) ((a) + (b))
__device__ inline uint32_t rowMinU32(uint32_t v) { MCRT_ROW_REDUCE(v, MCRT_OP_MIN); return v; }
__device__ inline uint32_t rowMaxU32(uint32_t v) { MCRT_ROW_REDUCE(v, MCRT_OP_MAX); return v; }
__device__ inline uint32_t rowSumU32(uint32_t v) { MCRT_ROW_REDUCE(v, MCRT_OP_ADD); return v; }
// inclusive prefix sum inside the row (zeros shifted in)
__device__ inline uint32_t rowPrefixU32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    return v;
}
// the row's 16 bits of a wave ballot
__device__ inline uint32_t rowBallot(bool pred) { return (uint32_t)(waveBallot(pred) >> (__lane_id() & 48u)) & 0xFFFFu; }
// value of lane `src16` of the own row (ds_bpermute: an LDS-crossbar operation — used once or twice per step, not in reductions)
__device__ inline uint32_t rowShfl(uint32_t v, uint32_t src16) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((__lane_id() & 48u) | src16) << 2), (int)v);
}

__device__ inline uint32_t grpBin(double d2v, double scale) {
    const double b = d2v * scale;
    return b < (double)(kGrpHist - 1u) ? (uint32_t)b : kGrpHist - 1u;
}

// Coarse k-th key of every participating row at once (the rows' version of waveSelectBound): a search from the top bit of
// the keys' high words for the smallest prefix that keeps >= k entries, stopped early when a prefix keeps k .. k + slack;
// every entry <= that key stays (compacted, buffer order kept), the rest cannot be among the k nearest. `part`, `count`,
// `bound` are uniform within a row; all 64 lanes must call.
__device__ inline void groupSelectBound(const GroupKnnLds& W, uint32_t& count, uint32_t k, uint32_t slack, bool part, double& bound) {
    const uint32_t l16 = __lane_id() & 15u, below = (1u << l16) - 1u;
    uint32_t hi[kGrpRows], lo[kGrpRows], id[kGrpRows];
#pragma unroll
    for (uint32_t s = 0; s < kGrpRows; s++) {
        const uint32_t j = l16 + 16u * s;
        const bool valid = part && j < count;
        union { double d; uint32_t u[2]; } c;
        c.d = valid ? W.d2[j] : 0.0;
        hi[s] = valid ? c.u[1] : 0xFFFFFFFFu;  // above every trial value (bit 31 of a trial is clear)
        lo[s] = c.u[0];
        id[s] = valid ? W.idx[j] : 0xFFFFFFFFu;
    }
    uint32_t Th = 0u;
    bool settled = !part, early = false;
    for (int bit = 30; bit >= 32 - kCoarseBits; bit--) {
        if (!waveBallot(!settled)) break;
        const uint32_t trial = Th | ((1u << bit) - 1u);
        uint32_t c = 0;
#pragma unroll
        for (uint32_t s = 0; s < kGrpRows; s++) c += hi[s] <= trial ? 1u : 0u;
        const uint32_t n_le = rowSumU32(c);
        if (!settled) {
            if (n_le < k) {
                Th |= (1u << bit);
            } else if (n_le <= k + slack) {
                Th = trial;
                settled = true;
                early = true;
            }
        }
    }
    if (part && !early) Th |= (1u << (32 - kCoarseBits)) - 1u;
    uint32_t out = 0;
#pragma unroll
    for (uint32_t s = 0; s < kGrpRows; s++) {
        const bool keep = part && hi[s] <= Th;
        const uint32_t gm = rowBallot(keep);
        if (keep) {
            const uint32_t slot = out + (uint32_t)__popc(gm & below);  // slot <= j, and every entry is in registers already
            union { double d; uint32_t u[2]; } c;
            c.u[1] = hi[s];
            c.u[0] = lo[s];
            W.d2[slot] = c.d;
            W.idx[slot] = id[s];
        }
        out += (uint32_t)__popc(gm);
    }
    if (part) {
        count = out;
        union { double d; uint32_t u[2]; } c;
        c.u[1] = Th;
        c.u[0] = 0xFFFFFFFFu;
        bound = c.d;
    }
}

// The k smallest of a row's `count` entries when count - k is small: the largest entry is dropped until k are left (among
// equal keys the one latest in the buffer first). Result compacted; r2 = the largest distance kept. All lanes must call.
__device__ inline void groupTrimToK(const GroupKnnLds& W, uint32_t& count, uint32_t k, bool part, double& r2) {
    const uint32_t l16 = __lane_id() & 15u, below = (1u << l16) - 1u;
    constexpr uint32_t kRows = 5;  // count <= k + 6 <= 70 < 80
    uint32_t hi[kRows], lo[kRows], id[kRows];
    bool valid[kRows];
#pragma unroll
    for (uint32_t s = 0; s < kRows; s++) {
        const uint32_t j = l16 + 16u * s;
        valid[s] = part && j < count;
        union { double d; uint32_t u[2]; } c;
        c.d = valid[s] ? W.d2[j] : 0.0;
        hi[s] = c.u[1];
        lo[s] = c.u[0];
        id[s] = valid[s] ? W.idx[j] : 0xFFFFFFFFu;
    }
    auto largest = [&](uint32_t& mh, uint32_t& ml) {  // keys are non-negative doubles: they order like integers
        uint32_t h = 0u;
#pragma unroll
        for (uint32_t s = 0; s < kRows; s++) h = valid[s] && hi[s] > h ? hi[s] : h;
        mh = rowMaxU32(h);
        uint32_t l = 0u;
#pragma unroll
        for (uint32_t s = 0; s < kRows; s++) l = valid[s] && hi[s] == mh && lo[s] > l ? lo[s] : l;
        ml = rowMaxU32(l);
    };
    uint32_t n = count;
    bool changed = false;
    while (waveBallot(part && n > k)) {
        uint32_t mh, ml;
        largest(mh, ml);
        // the owner latest in the buffer: largest j = l16 + 16 s
        uint32_t mine = 0u;
#pragma unroll
        for (uint32_t s = 0; s < kRows; s++)
            if (valid[s] && hi[s] == mh && lo[s] == ml) mine = ((s << 4) | l16) + 1u;
        const uint32_t last = rowMaxU32(mine);
        if (part && n > k) {
            if (mine == last && mine != 0u) {
                const uint32_t s_drop = (mine - 1u) >> 4;
#pragma unroll
                for (uint32_t s = 0; s < kRows; s++)
                    if (s == s_drop) valid[s] = false;
            }
            n--;
            changed = true;
        }
    }
    uint32_t mh, ml;
    largest(mh, ml);
    uint32_t out = 0;
#pragma unroll
    for (uint32_t s = 0; s < kRows; s++) {
        const uint32_t gm = rowBallot(valid[s]);
        if (valid[s] && changed) {
            const uint32_t slot = out + (uint32_t)__popc(gm & below);
            union { double d; uint32_t u[2]; } c;
            c.u[1] = hi[s];
            c.u[0] = lo[s];
            W.d2[slot] = c.d;
            W.idx[slot] = id[s];
        }
        out += (uint32_t)__popc(gm);
    }
    if (part) {
        count = n;
        union { double d; uint32_t u[2]; } c;
        c.u[1] = mh;
        c.u[0] = ml;
        r2 = n ? c.d : 0.0;
    }
}

// k-NN of p (uniform within a row; `on`: the row has a query). On return the row's buffer holds its result (unordered) in
// slots [0, n) and r2 the largest of its distances; `redo`: the row gave up (frontier full, or a crowd at the k-th distance)
// and the caller repeats the query with waveKnnSearch. All 64 lanes must call; k <= kGrpMaxK.
__device__ inline void groupKnnSearch(const PhotonMapViewW& map, d3 p, bool on, uint32_t k, const GroupKnnLds& W, uint32_t& n_out, double& r2_out,
                                      bool& redo, uint32_t& octant_visits) {
    n_out = 0u;
    r2_out = 0.0;
    redo = false;
    const PhotonMapView& m = map.base;
    if (m.num_octants == 0) return;
    if ((uint64_t)k > m.num_photons) k = (uint32_t)m.num_photons;
    if (k == 0) return;
    const uint32_t l16 = __lane_id() & 15u, below = (1u << l16) - 1u;
    constexpr uint32_t kNone = 0xFFFFFFFFu, kScan = 0x80000000u;
    float f_d2[kGrpFront];
    uint32_t f_a[kGrpFront], f_b[kGrpFront];  // b == 0: free slot
#pragma unroll
    for (uint32_t s = 0; s < kGrpFront; s++) {
        f_d2[s] = INFINITY;
        f_a[s] = kNone;
        f_b[s] = 0u;
    }
    double maxd2 = kDblMax, hscale = 0.0, hinv = 0.0;
    uint32_t count = 0u;
    bool dirty = false, bounded = false, hist_on = false;
    uint32_t cur_a = map.root_a, cur_b = map.root_m;
    bool act = on;
    auto histRebuild = [&](bool which) {  // the histogram follows the buffer (rows in `which`)
        if (which) {
#pragma unroll
            for (uint32_t b = 0; b < 4u; b++) W.hist[4u * l16 + b] = 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t s = 0; s < kGrpRows; s++) {
            const uint32_t j = l16 + 16u * s;
            if (which && j < count) __atomic_fetch_add(W.hist + grpBin(W.d2[j], hscale), 1u, __ATOMIC_RELAXED);
        }
    };
    while (waveBallot(act)) {
        if (act && l16 == 0u) octant_visits++;
        const bool inner = act && !(cur_b & kScan), scan = act && (cur_b & kScan) != 0u;
        if (waveBallot(inner)) {
            // ---- records of the octant (children / grandchildren): 16 per pass
            float cd2[4], corner = INFINITY;
            uint32_t ca[4], cb[4];
            bool push[4];
#pragma unroll
            for (uint32_t r = 0; r < 4u; r++) {
                const uint32_t rec = 16u * r + l16;
                cd2[r] = INFINITY;
                ca[r] = kNone;
                cb[r] = 0u;
                push[r] = false;
                if (inner && rec < cur_b) {
                    const WideRec* cr = map.wide + (size_t)cur_a + rec;
                    double bb[6];
                    for (int c = 0; c < 6; c++) bb[c] = cr->b[c];
                    const uint32_t rec_contained = cr->contained;
                    ca[r] = cr->a;
                    cb[r] = cr->m;
                    const double d2c = boxDistance2(bb, p);
                    push[r] = d2c <= maxd2;
                    cd2[r] = floatBelow(d2c);
                    // linear-octree.cpp:96-100; rounded UP to float: still an upper bound of the k-th distance
                    if (push[r] && rec_contained >= k) corner = fminf(corner, floatAbove(boxMaxDistance2(bb, p)));
                }
            }
            const double best_corner = (double)bitsFloat(rowMinU32(floatBits(corner)));
            if (inner && best_corner < maxd2) maxd2 = best_corner;
#pragma unroll
            for (uint32_t r = 0; r < 4u; r++) {
                // the bound of THIS step already applies to its records (linear-octree.cpp:91-101)
                push[r] = push[r] && (double)cd2[r] <= maxd2;
#pragma unroll
                for (uint32_t s = 0; s < kGrpFront; s++)
                    if (push[r] && f_b[s] == 0u) {
                        f_d2[s] = cd2[r];
                        f_a[s] = ca[r];
                        f_b[s] = cb[r];
                        push[r] = false;
                    }
                // (rare) all slots of the tester taken: any free slot of the row, one record per row and round
                while (waveBallot(push[r])) {
                    const uint32_t gm = rowBallot(push[r]);
                    const bool has = gm != 0u;
                    const uint32_t src16 = has ? (uint32_t)__ffs((int)gm) - 1u : 0u;
                    const float d = bitsFloat(rowShfl(floatBits(cd2[r]), src16));
                    const uint32_t a = rowShfl(ca[r], src16), b = rowShfl(cb[r], src16);
                    bool placed = !has;
#pragma unroll
                    for (uint32_t s = 0; s < kGrpFront; s++) {
                        const uint32_t fm = rowBallot(!placed && f_b[s] == 0u);
                        if (!placed && fm != 0u) {
                            if (l16 == (uint32_t)__ffs((int)fm) - 1u) {
                                f_d2[s] = d;
                                f_a[s] = a;
                                f_b[s] = b;
                            }
                            placed = true;
                        }
                    }
                    if (has && !placed) redo = true;  // 64 octants pending: this query goes to the wave search
                    if (has && l16 == src16) push[r] = false;
                }
            }
        }
        if (waveBallot(scan)) {
            // ---- photons of the octant: 16 per pass, four passes per round trip
            const uint32_t start = scan ? cur_a : 0u, contained = scan ? cur_b & ~kScan : 0u;
            if (scan && !hist_on && count == 0u && maxd2 < kDblMax) {  // first scan under a finite bound: the histogram spans [0, bound]
                hist_on = true;
                hscale = (double)kGrpHist / maxd2;
                hinv = maxd2 / (double)kGrpHist;
#pragma unroll
                for (uint32_t b = 0; b < 4u; b++) W.hist[4u * l16 + b] = 0u;
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t base = 0; waveBallot(base < contained); base += 64u) {
                float px[4], py[4], pz[4];
#pragma unroll
                for (uint32_t c = 0; c < 4u; c++) {
                    const uint32_t i = base + 16u * c + l16;
                    const uint32_t ii = i < contained ? i : (contained ? contained - 1u : 0u);  // clamp: keeps the loads unconditional
                    const float* ph = m.photons + (size_t)(start + ii) * 8;
                    px[c] = ph[3];
                    py[c] = ph[4];
                    pz[c] = ph[5];
                }
#pragma unroll
                for (uint32_t c = 0; c < 4u; c++) {
                    const uint32_t i = base + 16u * c + l16;
                    const d3 d = p - d3{(double)px[c], (double)py[c], (double)pz[c]};  // glm::distance2(data.pos(), p)
                    const double d2v = dot(d, d);
                    const bool cand = i < contained && d2v <= maxd2;
                    const unsigned long long wm = waveBallot(cand);
                    if (!wm) continue;
                    const uint32_t gm = (uint32_t)(wm >> (__lane_id() & 48u)) & 0xFFFFu;
                    if (cand) {
                        const uint32_t slot = count + (uint32_t)__popc(gm & below);
                        W.d2[slot] = d2v;
                        W.idx[slot] = start + i;
                        if (hist_on) __atomic_fetch_add(W.hist + grpBin(d2v, hscale), 1u, __ATOMIC_RELAXED);
                    }
                    count += (uint32_t)__popc(gm);
                    dirty = dirty || gm != 0u;
                    const bool full = count > kGrpCand - 16u;  // make room: drop what cannot be among the k nearest
                    if (waveBallot(full)) {
                        double bound = 0.0;
                        groupSelectBound(W, count, k, 4u, full, bound);
                        if (full) {
                            dirty = false;
                            bounded = true;
                            maxd2 = gmin(maxd2, bound);
                            if (count > kGrpCand - 16u) {  // a crowd inside 0.4 % of the k-th distance: the wave search sorts it out
                                redo = true;
                                count = 0u;
                            }
                        }
                        if (waveBallot(full && hist_on)) histRebuild(full && hist_on);
                    }
                }
            }
            // The k-th best so far bounds the answer (linear-octree.cpp:79). With the histogram: after every scan that added
            // candidates, the upper edge of the bin that holds the k-th nearest so far; without (no finite bound when the
            // scanning started): once, by the key search, as soon as k candidates exist.
            const bool need_h = scan && hist_on && dirty && count >= k;
            if (waveBallot(need_h)) {
                __builtin_amdgcn_wave_barrier();
                uint32_t h[4] = {0u, 0u, 0u, 0u};
                if (need_h) {
#pragma unroll
                    for (uint32_t b = 0; b < 4u; b++) h[b] = W.hist[4u * l16 + b];
                }
                const uint32_t sum = h[0] + h[1] + h[2] + h[3];
                const uint32_t incl = rowPrefixU32(sum), excl = incl - sum;
                if (need_h && l16 == 0u) W.mail[0] = kDblMax;
                __builtin_amdgcn_wave_barrier();
                if (need_h && incl >= k && excl < k) {  // exactly one lane of the row, if any
                    uint32_t running = excl, bin = 4u * l16 + 3u;
                    bool found = false;
#pragma unroll
                    for (uint32_t b = 0; b < 4u; b++) {
                        running += h[b];
                        if (!found && running >= k) {
                            bin = 4u * l16 + b;
                            found = true;
                        }
                    }
                    W.mail[0] = (double)(bin + 1u) * hinv * 1.000000000001;  // (an entry of this bin may sit a rounding above the exact edge)
                }
                __builtin_amdgcn_wave_barrier();
                if (need_h) {
                    const double edge = W.mail[0];
                    if (edge < kDblMax) {
                        maxd2 = gmin(maxd2, edge);
                        bounded = true;
                    }
                    dirty = false;
                }
            }
            const bool need_s = scan && !hist_on && dirty && count >= k && !bounded;
            if (waveBallot(need_s)) {
                double bound = 0.0;
                groupSelectBound(W, count, k, 4u, need_s, bound);
                if (need_s) {
                    dirty = false;
                    bounded = true;
                    maxd2 = gmin(maxd2, bound);
                }
            }
        }
        if (redo) act = false;
        // ---- pop the nearest octant of the row's frontier
        float mine = f_d2[0];
        uint32_t which = 0u;
#pragma unroll
        for (uint32_t s = 1; s < kGrpFront; s++)
            if (f_d2[s] < mine) {
                mine = f_d2[s];
                which = s;
            }
        if (!act) mine = INFINITY;
        const float best = bitsFloat(rowMinU32(floatBits(mine)));  // non-negative floats order like their bit patterns
        const bool go = act && best < INFINITY && !((double)best > maxd2);  // frontier empty / linear-octree.cpp:113
        const uint32_t gm = rowBallot(go && mine == best);
        const uint32_t src16 = gm ? (uint32_t)__ffs((int)gm) - 1u : 0u;
        uint32_t sel_a = f_a[0], sel_b = f_b[0];
#pragma unroll
        for (uint32_t s = 1; s < kGrpFront; s++)
            if (which == s) {
                sel_a = f_a[s];
                sel_b = f_b[s];
            }
        const uint32_t na = rowShfl(sel_a, src16), nb = rowShfl(sel_b, src16);
        if (go) {
            cur_a = na;
            cur_b = nb;
            if (l16 == src16) {
#pragma unroll
                for (uint32_t s = 0; s < kGrpFront; s++)
                    if (which == s) {
                        f_d2[s] = INFINITY;
                        f_b[s] = 0u;
                    }
            }
        }
        act = go;
    }
    // ---- exact selection, once: shrink to the entries that can still matter, then drop the largest until k are left
    const bool sel = on && !redo && count > k;
    if (waveBallot(sel)) {
        double bound = 0.0;
        groupSelectBound(W, count, k, 2u, sel, bound);
    }
    if (on && !redo && count > k + 6u) redo = true;  // a crowd at the k-th distance: the general selection of the wave search
    double r2 = 0.0;
    groupTrimToK(W, count, k, on && !redo, r2);
    if (on && !redo) {
        n_out = count;
        r2_out = r2;
    }
}

}  // namespace mcrt

#endif  // __HIPCC__
#endif
