// Wave-cooperative photon-map radiance estimate (the §8 rows a23/a24) — device only.
//
//   LinearOctree<Photon>::knnSearch          octree/linear-octree.cpp:25-117
//   PhotonMapper::estimate{Global,Caustic}Radiance   integrator/photon-mapper/photon-mapper.cpp:343-391
//
// A per-lane k-NN search keeps a k-entry heap and a frontier per lane; with k = 50 that is 600+ B per
// lane, which does not fit in LDS, and in global memory every heap operation is a chain of dependent
// DRAM/L2 round trips (measured: 21 M searches/s, slower than the CPU node). Here the 64 lanes of a
// wavefront serve ONE query at a time:
//   * the octree walk is wave-uniform; the frontier (octants still to visit) lives in registers, two
//     entries per lane; pop = wave-wide argmin; the <= 8 children of an octant are tested by 8 lanes in
//     parallel (child list precomputed at upload);
//   * a leaf (or any octant with <= k photons, linear-octree.cpp:51) is scanned 64 photons at a time
//     with coalesced 32-byte loads; photons within the current bound are appended to a per-wave
//     candidate buffer in LDS by ballot/prefix compaction; when the buffer fills, the exact k smallest
//     are kept (rank by counting) and the bound tightens to the k-th distance;
//   * the pruning rules are the reference's (inclusive <=, "an octant holding >= k photons bounds the
//     answer by its farthest corner"), so the k-set equals the reference's (ties at the k-th distance
//     aside, which the reference resolves by insertion order);
//   * the k photons are then evaluated by k lanes in parallel (BSDF towards the photon direction) with
//     the query lane's Interaction broadcast by shuffles, and summed by a wave reduction. The order of
//     that FP64 sum differs from the reference's heap-array order: a few ulp.
#pragma once

#include "mcrt_integrator.hpp"
#include "mcrt_lanesm.hpp"

#if defined(__HIPCC__) || defined(MCRT_WAVE_EMU)  // (MCRT_WAVE_EMU: tests/emu/wave_emu.hpp runs this file on the host, 64 fibers per wave)

// Where one lane reads what ANOTHER lane of its wave wrote to LDS a few instructions earlier, with no cross-lane operation in between,
// the code says so. On the device that is a wavefront-scope fence plus the compiler's wave barrier: LDS operations of one wave execute in
// order, so no instruction is needed (tools/compare_device_code.py: the kernels' instructions are unchanged), but the hand-off is now
// inside the memory model - the compiler may not move the write below or the read above this point. On the host
// (tests/emu/wave_emu.hpp) it is a rendezvous of the 64 fibers, where a lane runs its whole stretch between two cross-lane
// operations at once.
#if defined(MCRT_WAVE_EMU)
#define MCRT_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define MCRT_LOCKSTEP()                                      \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

namespace mcrt {

// The candidate buffer of a wave holds R rows of 64 entries (template parameter of everything below that keeps the buffer in
// registers row by row). R = 4 serves k <= 128 — every scene of the reference asks for 50; R = 16 serves k <= 768 (round 4: larger k
// used to fall to the per-lane legacy kernel); only beyond that does the per-lane kernel still run.
constexpr int kWaveRows = 4, kWaveRowsLarge = 16;
__host__ __device__ constexpr uint32_t waveCand(int R) { return 64u * (uint32_t)R; }  // entries per wave (pruned when > waveCand - 64)
__host__ __device__ constexpr uint32_t waveMaxK(int R) { return R <= 4 ? 128u : waveCand(R) - 256u; }
constexpr uint32_t kWaveCand = waveCand(kWaveRows);

constexpr uint32_t kWaveHist = 128;   // bins of the per-wave distance histogram (see "Selection by histogram")
// what a search sets its overflow word to: the photon-mapping megakernel adds the word into the SAME statistics word as the lanes'
// traversal-stack overflows (one per lane), and the host tells the two apart by the bits above 15 (mcrt_render_finish)
constexpr uint32_t kKnnOverflowFlag = 0x10000u;
static_assert(kKnnOverflowFlag == kLaneKnnOverflow, "one flag for the wave-cooperative and the per-lane searches");
constexpr uint32_t kWaveStateBytes = 16u;  // per wave, for the searches that keep their spill list's state in LDS ("Frontier overflow" below)
__host__ __device__ constexpr uint32_t waveKnnBytes(int R) { return waveCand(R) * 12u + kWaveHist * 4u; }  // LDS per wave: candidates + histogram
constexpr uint32_t kWaveKnnBytes = waveKnnBytes(kWaveRows);

struct WaveKnnLds {
    MCRT_LDS_AS double* d2;      // [waveCand(R)] this wave's candidate distances
    MCRT_LDS_AS uint32_t* idx;   // [waveCand(R)] photon indices
    MCRT_LDS_AS uint32_t* hist;  // [kWaveHist] candidates per distance2 bin
    uint32_t* spill = nullptr;   // the spill list's address for the searches that keep it in registers (kSpillRegs, below)
    MCRT_LDS_AS uint32_t* state = nullptr;  // [4] ... or, for the others, where its count and address are kept in LDS
};
// Frontier overflow. The reference's frontier is an unbounded priority queue (linear-octree.cpp:33); here it is 2 entries per lane in
// registers. An octree whose leaves hold far fewer photons than k can have more octants than that within the bound at once (k = 300
// on leaves of <= 200: seen). Entries that find no free slot go to a list in memory (lane 0 writes them); whenever the register
// frontier runs empty - or holds only octants beyond the bound - the list is read back (entries beyond the bound dropped). Visiting
// order only matters for speed: every octant within the final bound is visited, so the k-set is the same. A wave without a list
// (null address), or a list that overflows too, raises the overflow flag as before. Where the list's count and address live is a
// template choice (kSpillRegs), because the two eye-pass kernels answer it differently (measured, one box): in registers the kernel that
// walks a tree in memory (C5) spills 110 more VGPRs and loses 1.7 % (788 vs 775 ms at 64 spp); in LDS behind the histogram - touched
// only when an entry really overflows, the loop carries one flag - the kernel of LDS-resident scenes loses 8 % (hexagon_room 106 vs 98
// ms - and so does moving the histograms of its waves 16 bytes apart to make room). So: registers for LDS-resident scenes and the
// operator kernels; LDS, in a small array of its own behind the waves' buffers, for trees in memory.
constexpr uint32_t kWaveSpill = 1024;
// every wave, once, before its first search
__device__ inline void waveKnnInit(const WaveKnnLds& W, uint32_t* spill) {
    if (__lane_id() == 0) {
        const unsigned long long u = (unsigned long long)spill;
        W.state[0] = 0u;
        W.state[1] = (uint32_t)u;
        W.state[2] = (uint32_t)(u >> 32);
    }
    __builtin_amdgcn_wave_barrier();
}
// entries in the list (wave-uniform) and its address
__device__ inline uint32_t waveSpillState(const WaveKnnLds& W, uint32_t*& list) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.state[0]);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.state[1]);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.state[2]);
    list = (uint32_t*)(((unsigned long long)hi << 32) | lo);
    return n;
}
__device__ inline void waveSpillSetCount(const WaveKnnLds& W, uint32_t n) {
    if (__lane_id() == 0) W.state[0] = n;
    __builtin_amdgcn_wave_barrier();
}
// one more entry {distance2 as float bits, a, b} (wave-uniform values); false: no list, or full. All lanes must call.
__device__ inline bool waveSpillPush(const WaveKnnLds& W, float d, uint32_t a, uint32_t b) {
    uint32_t* list;
    const uint32_t n = waveSpillState(W, list);
    if (!list || n >= kWaveSpill) return false;
    if (__lane_id() == 0) {
        uint32_t* e = list + 3u * n;
        __hip_atomic_store(e + 0, __float_as_uint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(e + 1, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(e + 2, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    waveSpillSetCount(W, n + 1u);
    return true;
}

// What one step of the descent reads: for an inner octant, up to 64 records — its children that can be scanned right
// away (leaves, or octants with <= k photons, linear-octree.cpp:51) and, for every other child, that child's children
// instead (two octree levels per step: half as many dependent steps, and the box tests use the whole wave instead of 8
// lanes). Skipping the boxes of the expanded children only loosens pruning (a grandchild's box lies inside its parent's).
// One 64-byte record per entry, the records of an octant contiguous.
struct WideRec {
    double b[6];
    uint32_t contained;
    uint32_t a;      // scannable entry: first photon; inner entry: first record of ITS list
    uint32_t m;      // scannable entry: 0x80000000 | photon count; inner entry: number of records in its list (1..64)
    uint32_t pad;
};

// A photon's position alone, 12 bytes: what a leaf scan reads. The reference's 32-byte Photon (photon.hpp:36-37: flux, position,
// two angles) is only needed of the k photons an estimate ends up with; scanning the records themselves moved 32 bytes per
// photon for the 12 it looked at (round 4: a copy of the positions, made when a map is built or uploaded).
struct PhotonPos {
    float x, y, z;
};

struct PhotonMapViewW {
    PhotonMapView base;
    const WideRec* wide;
    uint32_t root_a, root_m;  // the root as an entry
    const PhotonPos* pos;     // [num_photons], in map order
};

// Broadcast from lane `src` (wave-uniform): v_readlane, no LDS round trip.
__device__ inline double waveShflD(double v, int src) {
    union { double d; int u[2]; } c;
    c.d = v;
    c.u[0] = __builtin_amdgcn_readlane(c.u[0], src);
    c.u[1] = __builtin_amdgcn_readlane(c.u[1], src);
    return c.d;
}
__device__ inline d3 waveShfl3(d3 v, int src) { return d3{waveShflD(v.x, src), waveShflD(v.y, src), waveShflD(v.z, src)}; }

__device__ inline double waveMinD(double v) {
    for (int off = 32; off > 0; off >>= 1) {
        union { double d; unsigned u[2]; } c;
        c.d = v;
        c.u[0] = __shfl_xor(c.u[0], off, 64);
        c.u[1] = __shfl_xor(c.u[1], off, 64);
        v = c.d < v ? c.d : v;
    }
    return v;
}
// Wave-wide FP64 sum through the DPP cross-lane network (pairs in quads, half rows, rows, then the four rows into
// lane 63). Lanes outside a step's row mask add 0.
#define MCRT_DPP_ADD_F64(v, ctrl, row_mask)                                               \
    do {                                                                                  \
        union { double d; int u[2]; } a_, b_;                                             \
        a_.d = (v);                                                                       \
        b_.u[0] = __builtin_amdgcn_update_dpp(0, a_.u[0], ctrl, row_mask, 0xF, false);    \
        b_.u[1] = __builtin_amdgcn_update_dpp(0, a_.u[1], ctrl, row_mask, 0xF, false);    \
        (v) = (v) + b_.d;                                                                 \
    } while (0)
__device__ inline double waveSumD(double v) {
    MCRT_DPP_ADD_F64(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    MCRT_DPP_ADD_F64(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    MCRT_DPP_ADD_F64(v, 0x141, 0xF);  // row_half_mirror
    MCRT_DPP_ADD_F64(v, 0x140, 0xF);  // row_mirror
    MCRT_DPP_ADD_F64(v, 0x142, 0xA);  // row_bcast15 -> rows 1, 3
    MCRT_DPP_ADD_F64(v, 0x143, 0xC);  // row_bcast31 -> rows 2, 3
    union { double d; int u[2]; } c;
    c.d = v;
    c.u[0] = __builtin_amdgcn_readlane(c.u[0], 63);
    c.u[1] = __builtin_amdgcn_readlane(c.u[1], 63);
    return c.d;
}

// Wave-wide minimum of a u32 through the DPP cross-lane network (no LDS round trips, unlike __shfl): swap within
// quads, mirror within half rows and rows, then fold the four rows of 16 into lane 63.
__device__ inline uint32_t waveMinU32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
    v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline uint32_t waveMaxU32(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false);
    v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false);
    v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Non-negative floats order like their bit patterns.
__device__ inline float waveMinPosF(float v) { return bitsFloat(waveMinU32(floatBits(v))); }
__device__ inline uint32_t waveRead(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }  // src wave-uniform

// Keep the k smallest of the first `count` buffer entries, compacted (unordered) into slots [0, k);
// returns min(count, k) and, in kth_d2, the largest distance kept. Exact selection by a bitwise radix
// search for the k-th smallest key: squared distances are non-negative doubles, so their bit patterns
// order like unsigned integers; 64 wave-uniform steps of "how many keys are <= prefix" (ballot +
// popcount) instead of an all-pairs ranking. Entries equal to the k-th key are kept in buffer order
// until k are reached (the reference resolves such ties by insertion order, linear-octree.cpp:58-77).
// All lanes must call.
template <int R = kWaveRows>
__device__ inline uint32_t waveSelectK(const WaveKnnLds& W, uint32_t count, uint32_t k, double& kth_d2) {
    const uint32_t lane = __lane_id();
    MCRT_LOCKSTEP();  // (the buffer was just written by other lanes)
    const int rows = (int)((count + 63u) / 64u);  // buffer rows in use (wave-uniform)
    unsigned long long key[R];
    uint32_t my_i[R];
    bool valid[R];
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        const uint32_t j = lane + 64u * s;
        valid[s] = j < count;
        union { double d; unsigned long long u; } c;
        c.d = valid[s] ? W.d2[j] : 0.0;
        key[s] = c.u;
        my_i[s] = valid[s] ? W.idx[j] : 0xFFFFFFFFu;
    }
    if (count <= k) {  // nothing to drop: only the largest distance is needed
        double mx = 0.0;
        _Pragma("unroll") for (int s = 0; s < R; s++) {
            union { double d; unsigned long long u; } c;
            c.u = key[s];
            if (valid[s] && c.d > mx) mx = c.d;
        }
        kth_d2 = -waveMinD(-mx);
        return count;
    }
    // T = the k-th smallest key: smallest T with #{key <= T} >= k, built from the top bit down
    unsigned long long T = 0ull;
    for (int bit = 62; bit >= 0; bit--) {  // bit 63 (sign) is clear in every key
        const unsigned long long trial = T | ((1ull << bit) - 1ull);  // all candidates with this bit clear
        uint32_t n_le = 0;
        _Pragma("unroll") for (int s = 0; s < R; s++)
            if (s < rows) n_le += __popcll(waveBallot(valid[s] && key[s] <= trial));
        if (n_le < k) T |= (1ull << bit);
    }
    uint32_t n_lt = 0;
    _Pragma("unroll") for (int s = 0; s < R; s++) n_lt += __popcll(waveBallot(valid[s] && key[s] < T));
    // compaction: everything below T, then entries equal to T until k are kept
    uint32_t out = 0, eq_left = k - n_lt;
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        const bool lt = valid[s] && key[s] < T;
        const bool eq = valid[s] && key[s] == T;
        const unsigned long long m_lt = waveBallot(lt), m_eq = waveBallot(eq);
        const uint32_t eq_rank = __popcll(m_eq & ((1ull << lane) - 1ull));
        const bool keep_eq = eq && eq_rank < eq_left;
        const unsigned long long m_keep = m_lt | waveBallot(keep_eq);
        const bool keep = lt || keep_eq;
        if (keep) {
            const uint32_t slot = out + __popcll(m_keep & ((1ull << lane) - 1ull));
            union { double d; unsigned long long u; } c;
            c.u = key[s];
            W.d2[slot] = c.d;
            W.idx[slot] = my_i[s];
        }
        out += __popcll(m_keep);
        const uint32_t n_eq_kept = __popcll(waveBallot(keep_eq));
        eq_left -= n_eq_kept;
    }
    union { double d; unsigned long long u; } c;
    c.u = T;
    kth_d2 = c.d;
    return k;
}

// Pruning only needs AN upper bound of the k-th smallest distance, and room in the buffer: the same radix search
// stopped after the top `kCoarseBits` bits gives T = (prefix of the k-th key) | (all lower bits set) >= k-th key.
// Every entry <= T is kept (at least k, plus the few that share the k-th key's prefix: 8 mantissa bits = 0.4 %
// in distance2), the rest can never be among the k nearest. 19 wave-uniform steps instead of 63; the exact
// selection runs once, at the end of the search. Returns the new count; all lanes must call.
constexpr int kCoarseBits = 20;
// `slack`: the search for the bits stops as soon as a prefix keeps between k and k + slack entries.
template <int R = kWaveRows>
__device__ inline uint32_t waveSelectBound(const WaveKnnLds& W, uint32_t count, uint32_t k, double& bound_d2, uint32_t slack = 0) {
    const uint32_t lane = __lane_id();
    MCRT_LOCKSTEP();  // (the buffer was just written by other lanes)
    const int rows = (int)((count + 63u) / 64u);  // buffer rows in use (wave-uniform)
    uint32_t hi[R], lo[R], my_i[R];
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        const uint32_t j = lane + 64u * s;
        const bool valid = j < count;
        union { double d; uint32_t u[2]; } c;
        c.d = valid ? W.d2[j] : 0.0;
        hi[s] = valid ? c.u[1] : 0xFFFFFFFFu;
        lo[s] = c.u[0];
        my_i[s] = valid ? W.idx[j] : 0xFFFFFFFFu;
    }
    // The bits searched (62 .. 44) all lie in the keys' high words and every trial value has all lower bits set, so
    // key <= trial is hi(key) <= hi(trial): 32-bit compares (full rate; the 64-bit integer ones are not) and 32-bit scalar
    // arithmetic in the chain from one step to the next. Invalid entries carry the largest high word.
    uint32_t Th = 0u;
    bool settled = false;
    for (int bit = 30; bit >= 32 - kCoarseBits; bit--) {  // bit 31 of the high word (the sign) is clear in every key
        const uint32_t trial = Th | ((1u << bit) - 1u);
        uint32_t n_le = 0;
        _Pragma("unroll") for (int s = 0; s < R; s++)
            if (s < rows) n_le += __popcll(waveBallot(hi[s] <= trial));
        if (n_le < k) {
            Th |= (1u << bit);
        } else if (n_le <= k + slack) {  // k .. k + slack entries below this trial value: good enough a bound
            Th = trial;
            settled = true;
            break;
        }
    }
    // (two bits per step — three trial values counted side by side, half the dependent steps — measured slower: 196 -> 180 M/s)
    if (!settled) Th |= (1u << (32 - kCoarseBits)) - 1u;
    uint32_t out = 0;
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        const bool keep = hi[s] <= Th;  // (an invalid entry's high word is above every Th: bit 31 of Th is clear)
        const unsigned long long m_keep = waveBallot(keep);
        if (keep) {
            const uint32_t slot = out + __popcll(m_keep & ((1ull << lane) - 1ull));
            union { double d; uint32_t u[2]; } c;
            c.u[1] = hi[s];
            c.u[0] = lo[s];
            W.d2[slot] = c.d;
            W.idx[slot] = my_i[s];
        }
        out += __popcll(m_keep);
    }
    union { double d; uint32_t u[2]; } c;
    c.u[1] = Th;
    c.u[0] = 0xFFFFFFFFu;
    bound_d2 = c.d;
    return out;
}

// The k smallest of count <= 64 entries (count - k is small): the largest entry is dropped until k are left — among equal
// keys the one latest in the buffer first, which keeps the entries equal to the k-th key "in buffer order until k are
// reached" like waveSelectK. Result compacted into slots [0, n), kth_d2 = the largest distance kept. All lanes must call.
__device__ inline uint32_t waveTrimToK(const WaveKnnLds& W, uint32_t count, uint32_t k, double& kth_d2) {
    const uint32_t lane = __lane_id();
    MCRT_LOCKSTEP();  // (the buffer was just written by other lanes)
    bool valid = lane < count;
    union { double d; uint32_t u[2]; } c;
    c.d = valid ? W.d2[lane] : 0.0;
    const uint32_t my_i = valid ? W.idx[lane] : 0xFFFFFFFFu;
    const uint32_t hi = c.u[1], lo = c.u[0];
    auto largest = [&](uint32_t& mh, uint32_t& ml) {  // the largest valid key (keys are non-negative doubles: they order like integers)
        mh = waveMaxU32(valid ? hi : 0u);
        ml = waveMaxU32((valid && hi == mh) ? lo : 0u);
    };
    uint32_t n = count;
    while (n > k) {  // wave-uniform
        uint32_t mh, ml;
        largest(mh, ml);
        const unsigned long long owners = waveBallot(valid && hi == mh && lo == ml);
        const int drop = 63 - __clzll((long long)owners);
        if ((int)lane == drop) valid = false;
        n--;
    }
    uint32_t mh = 0u, ml = 0u;
    if (n > 0) largest(mh, ml);
    union { double d; uint32_t u[2]; } r;
    r.u[1] = mh;
    r.u[0] = ml;
    kth_d2 = r.d;
    if (n != count) {  // close the gaps
        const unsigned long long keep = waveBallot(valid);
        if (valid) {
            const uint32_t slot = __popcll(keep & ((1ull << lane) - 1ull));
            W.d2[slot] = c.d;
            W.idx[slot] = my_i;
        }
    }
    return n;
}

// Sort the first n result entries ascending by (distance2, index) in place (n <= 64 S).
template <int S = 2>
__device__ inline void waveSortResult(const WaveKnnLds& W, uint32_t n) {
    const uint32_t lane = __lane_id();
    MCRT_LOCKSTEP();  // (the buffer was just written by other lanes)
    double my_d[S];
    uint32_t my_i[S], rank[S];
    _Pragma("unroll") for (int s = 0; s < S; s++) {
        const uint32_t j = lane + 64u * s;
        my_d[s] = j < n ? W.d2[j] : INFINITY;
        my_i[s] = j < n ? W.idx[j] : 0xFFFFFFFFu;
        rank[s] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        const double d = W.d2[i];
        const uint32_t id = W.idx[i];
        _Pragma("unroll") for (int s = 0; s < S; s++) rank[s] += (d < my_d[s] || (d == my_d[s] && id < my_i[s])) ? 1u : 0u;
    }
    MCRT_LOCKSTEP();  // every lane has read the whole list before any lane writes
    _Pragma("unroll") for (int s = 0; s < S; s++) {
        const uint32_t j = lane + 64u * s;
        if (j < n) {
            W.d2[rank[s]] = my_d[s];
            W.idx[rank[s]] = my_i[s];
        }
    }
}

// ---- Selection by histogram. The two key searches of a query — the bound once k candidates exist, the exact k-set at
// the end — cost 12 500 of the 29 000 cycles a search takes: chains of dependent scalar instructions, 19 + 19 steps. For
// photons on surfaces the squared distances of the candidates are spread almost evenly over [0, R] (the count within
// radius r grows like r^2), so a histogram of distance2 over the bound R in force when scanning starts answers both:
// every candidate adds itself to its bin on the way into the buffer (one LDS atomic), the k-th nearest lies in the first
// bin whose running count reaches k (a wave prefix sum over 128 bins), that bin's upper edge bounds the search, and at
// the end everything in lower bins belongs to the result while only the boundary bin — a candidate or two — needs an
// exact selection. Entries beyond the buffer's last reduction are re-added when the buffer is reduced (rare).
struct WaveHist {
    bool on;          // a finite bound existed when scanning started: the histogram is in use
    double scale;     // bins per unit of distance2: kWaveHist / R
    double inv_scale;
};
__device__ inline uint32_t histBin(const WaveHist& H, double d2v) {
    const double b = d2v * H.scale;
    return b < (double)(kWaveHist - 1u) ? (uint32_t)b : kWaveHist - 1u;
}
__device__ inline void histClear(const WaveKnnLds& W) {
    const uint32_t lane = __lane_id();
    W.hist[lane] = 0u;
    W.hist[lane + 64u] = 0u;
}
__device__ inline void histAdd(const WaveKnnLds& W, const WaveHist& H, double d2v) { __atomic_fetch_add(W.hist + histBin(H, d2v), 1u, __ATOMIC_RELAXED); }
// inclusive prefix sum over the 64 lanes through the DPP network: shifts by 1, 2, 4, 8 inside the rows of 16 (zeros shifted
// in), then the last lane of row 0 / 2 onto rows 1 / 3 and the last lane of row 1 onto rows 2 and 3. (__shfl_up would be
// six LDS-crossbar permutes in a chain.)
__device__ inline uint32_t wavePrefixU32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2, 3
    return v;
}
// The bin in which the running count first reaches k: returns true and the bin, the count in lower bins and the count up
// to and including it; false when fewer than k candidates are in the histogram. All lanes must call.
__device__ inline bool histKthBin(const WaveKnnLds& W, uint32_t k, uint32_t& bin, uint32_t& below, uint32_t& upto) {
    const uint32_t lane = __lane_id();
    __builtin_amdgcn_wave_barrier();
    const uint32_t h0 = W.hist[2u * lane], h1 = W.hist[2u * lane + 1u];
    const uint32_t incl = wavePrefixU32(h0 + h1);  // through bin 2 lane + 1
    const unsigned long long reached = waveBallot(incl >= k);
    if (!reached) return false;
    const int l = __ffsll((long long)reached) - 1;
    const uint32_t incl_l = (uint32_t)__builtin_amdgcn_readlane((int)incl, l);
    const uint32_t h0_l = (uint32_t)__builtin_amdgcn_readlane((int)h0, l), h1_l = (uint32_t)__builtin_amdgcn_readlane((int)h1, l);
    const uint32_t before = incl_l - h0_l - h1_l;  // bins below 2 l
    if (before + h0_l >= k) {
        bin = 2u * (uint32_t)l;
        below = before;
        upto = before + h0_l;
    } else {
        bin = 2u * (uint32_t)l + 1u;
        below = before + h0_l;
        upto = incl_l;
    }
    return true;
}

// The exact k nearest from the histogram: all candidates of the bins below the k-th one's, and of the boundary bin the
// (k - below) smallest (the largest is dropped until that many are left — among equal keys the one latest in the buffer
// first). Result compacted into slots [0, k); kth_d2 = the largest distance kept. Falls back (returns 0) when the
// boundary bin holds too many candidates for that (the caller then runs the general selection). All lanes must call.
template <int R = kWaveRows>
__device__ inline uint32_t histSelectK(const WaveKnnLds& W, const WaveHist& H, uint32_t count, uint32_t k, double& kth_d2) {
    const uint32_t lane = __lane_id();
    uint32_t bin, below, upto;
    if (!histKthBin(W, k, bin, below, upto)) return 0u;
    const uint32_t need = k - below, have = upto - below;  // of the boundary bin
    if (have > 64u || have - need > 6u) return 0u;
    const int rows = (int)((count + 63u) / 64u);
    // pass 1: entries of lower bins keep their relative order at the front; boundary entries are gathered in registers (one per lane)
    double keep_d[R];
    uint32_t keep_i[R];
    bool low[R], edge[R];
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        const uint32_t j = lane + 64u * s;
        const bool valid = s < rows && j < count;
        keep_d[s] = valid ? W.d2[j] : 0.0;
        keep_i[s] = valid ? W.idx[j] : 0xFFFFFFFFu;
        const uint32_t b = valid ? histBin(H, keep_d[s]) : 0xFFFFFFFFu;
        low[s] = valid && b < bin;
        edge[s] = valid && b == bin;
    }
    // boundary entries -> lanes 0 .. have-1 (in buffer order) through LDS slots [kWaveCand - 64, kWaveCand) ... they may still hold
    // live entries, so use registers + a staging pass: first write the low entries, then stage the boundary ones behind them
    uint32_t out = 0;
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        if (s >= rows) break;
        const unsigned long long m = waveBallot(low[s]);
        if (low[s]) {
            const uint32_t slot = out + __popcll(m & ((1ull << lane) - 1ull));
            W.d2[slot] = keep_d[s];   // slot <= j: never overwrites an entry that has not been read (all reads are done above)
            W.idx[slot] = keep_i[s];
        }
        out += __popcll(m);
    }
    uint32_t eout = 0;
    _Pragma("unroll") for (int s = 0; s < R; s++) {
        if (s >= rows) break;
        const unsigned long long m = waveBallot(edge[s]);
        if (edge[s]) {
            const uint32_t slot = out + eout + __popcll(m & ((1ull << lane) - 1ull));
            W.d2[slot] = keep_d[s];
            W.idx[slot] = keep_i[s];
        }
        eout += __popcll(m);
    }
    // (out == below and eout == have by construction; the boundary entries now sit in slots [below, below + have))
    // pass 2: of the boundary entries keep the `need` smallest
    MCRT_LOCKSTEP();  // (staged by other lanes just above)
    bool valid = lane < eout;
    union { double d; uint32_t u[2]; } c;
    c.d = valid ? W.d2[out + lane] : 0.0;
    const uint32_t my_i = valid ? W.idx[out + lane] : 0xFFFFFFFFu;
    auto largest = [&](uint32_t& mh, uint32_t& ml) {
        mh = waveMaxU32(valid ? c.u[1] : 0u);
        ml = waveMaxU32((valid && c.u[1] == mh) ? c.u[0] : 0u);
    };
    uint32_t n = eout;
    while (n > need) {  // wave-uniform, at most 6 rounds
        uint32_t mh, ml;
        largest(mh, ml);
        const unsigned long long owners = waveBallot(valid && c.u[1] == mh && c.u[0] == ml);
        const int drop = 63 - __clzll((long long)owners);
        if ((int)lane == drop) valid = false;
        n--;
    }
    uint32_t mh = 0u, ml = 0u;
    largest(mh, ml);  // need >= 1: the k-th nearest is in the boundary bin
    union { double d; uint32_t u[2]; } r;
    r.u[1] = mh;
    r.u[0] = ml;
    kth_d2 = r.d;
    if (n != eout) {
        const unsigned long long keep = waveBallot(valid);
        if (valid) {
            const uint32_t slot = out + __popcll(keep & ((1ull << lane) - 1ull));
            W.d2[slot] = c.d;
            W.idx[slot] = my_i;
        }
    }
    return out + n;
}

// k-NN of point p (wave-uniform) in `map`. On return the buffer holds the result (unordered) in slots
// [0, n) and r2_max the largest of its distances; returns n. All 64 lanes must call with the same arguments.
//
// A frontier entry carries everything the visit of its octant needs — {distance2, a, b}: scannable octant (leaf,
// or <= k photons, linear-octree.cpp:51) a = first photon, b = 0x80000000 | count; inner octant a = first record of its
// list, b = number of records — so that popping an octant costs no memory access and a visit is ONE round trip (its
// record list, or its photons). Lane l tests record l and drops it, if kept, into its own two frontier slots; only when a
// lane's slots are both taken does the wave look for a free slot elsewhere.
//
// `bound2`: an upper bound of the k-th nearest photon's squared distance known beforehand (kDblMax: none). The search then
// starts pruning where it would otherwise arrive after its first scans; the k-set is the same (every one of the k nearest
// lies within ANY upper bound, and all comparisons against the bound are inclusive).
template <int R = kWaveRows, bool kSpillRegs = true>
__device__ inline uint32_t waveKnnSearch(const PhotonMapViewW& map, d3 p, uint32_t k, const WaveKnnLds& W, double& r2_max,
                                         uint32_t& overflow, uint32_t& octant_visits, double bound2 = kDblMax) {
    r2_max = 0.0;
    const PhotonMapView& m = map.base;
    if (m.num_octants == 0) return 0;
    if ((uint64_t)k > m.num_photons) k = (uint32_t)m.num_photons;
    if (k == 0) return 0;
    const uint32_t lane = __lane_id();
    constexpr uint32_t kNone = 0xFFFFFFFFu, kScan = 0x80000000u;
    // distances in the frontier are rounded DOWN to float: they only order the visits and end the search, and a
    // smaller key can only postpone the end (linear-octree.cpp:113 stays conservative)
    float f_d2[2] = {INFINITY, INFINITY};
    uint32_t f_a[2] = {kNone, kNone}, f_b[2] = {0u, 0u};  // b == 0 marks a free slot (an inner entry has b = 1)
    double max_distance2 = bound2;
    uint32_t count = 0;
    bool dirty = false;    // candidates appended since the buffer was last reduced
    bool bounded = false;  // max_distance2 has been tightened to a k-photon radius at least once
    WaveHist H;
    H.on = false;
    H.scale = H.inv_scale = 0.0;
    uint32_t cur_a = map.root_a, cur_b = map.root_m;  // root
    bool spilled = false;  // wave-uniform: this search has put entries on the wave's list in memory
    uint32_t sp_n = 0u;    // ... how many are on it now (kSpillRegs; otherwise read from LDS when `spilled`)
    for (;;) {
        octant_visits++;
        if (cur_b & kScan) {
            const uint32_t start = cur_a, contained = cur_b & ~kScan;
            if (!H.on && count == 0u && max_distance2 < kDblMax) {  // first scan under a finite bound: the histogram spans [0, bound]
                H.on = true;
                H.scale = (double)kWaveHist / max_distance2;
                H.inv_scale = max_distance2 / (double)kWaveHist;
                histClear(W);
            }
            // 4 x 64 photons per round trip: the four position loads of a lane are issued together
            for (uint32_t base = 0; base < contained; base += 256) {
                float px[4], py[4], pz[4];
                for (int c = 0; c < 4; c++) {
                    const uint32_t i = base + 64u * c + lane;
                    const uint32_t ii = i < contained ? i : contained - 1;  // clamp: keeps the loads unconditional
                    const PhotonPos q = map.pos[(size_t)(start + ii)];
                    px[c] = q.x;
                    py[c] = q.y;
                    pz[c] = q.z;
                }
                for (int c = 0; c < 4; c++) {
                    if (base + 64u * c >= contained) break;  // wave-uniform
                    const uint32_t i = base + 64u * c + lane;
                    d3 d = p - d3{(double)px[c], (double)py[c], (double)pz[c]};  // glm::distance2(data.pos(), p)
                    const double d2v = dot(d, d);
                    const bool cand = i < contained && d2v <= max_distance2;
                    const unsigned long long mask = waveBallot(cand);
                    if (mask) {
                        const uint32_t slot = count + __popcll(mask & ((1ull << lane) - 1ull));
                        if (cand) {
                            W.d2[slot] = d2v;
                            W.idx[slot] = start + i;
                            if (H.on) histAdd(W, H, d2v);
                        }
                        count += __popcll(mask);
                        dirty = true;
                        if (count > waveCand(R) - 64u) {  // make room: drop what cannot be among the k nearest
                            double bound;
                            count = waveSelectBound<R>(W, count, k, bound, 4u);
                            if (count > waveCand(R) - 64u) count = waveSelectK<R>(W, count, k, bound);  // a crowd inside 0.4 %
                            dirty = false;
                            bounded = true;
                            max_distance2 = gmin(max_distance2, bound);
                            if (H.on) {  // the histogram follows the buffer
                                histClear(W);
                                __builtin_amdgcn_wave_barrier();
                                for (uint32_t j = lane; j < count; j += 64u) histAdd(W, H, W.d2[j]);
                            }
                        }
                    }
                }
            }
            // The k-th best so far bounds the answer (linear-octree.cpp:79): taken once, as soon as k candidates
            // exist (it shrinks the bound from an octant diagonal to the k-photon radius); afterwards only when
            // the buffer fills, since every later candidate already lies within that radius.
            if (H.on) {
                // ... with the histogram after every scan that added candidates: the upper edge of the bin that holds the k-th
                // nearest so far (nothing is dropped from the buffer: later candidates are simply held to the tighter bound)
                if (dirty && count >= k) {
                    uint32_t bin, below, upto;
                    if (histKthBin(W, k, bin, below, upto)) {
                        const double edge = (double)(bin + 1u) * H.inv_scale * 1.000000000001;  // (an entry of this bin may sit a rounding above the exact edge)
                        max_distance2 = gmin(max_distance2, edge);
                        bounded = true;
                    }
                    dirty = false;
                }
            } else if (dirty && count >= k && !bounded) {
                double bound;
                count = waveSelectBound<R>(W, count, k, bound, 4u);
                dirty = false;
                bounded = true;
                max_distance2 = gmin(max_distance2, bound);
            }
        } else {
            // records of the octant (children / grandchildren): one per lane
            float cd2 = INFINITY, corner = INFINITY;
            uint32_t ca = kNone, cb = 0u;
            bool push = false;
            if (lane < cur_b) {
                const WideRec* cr = map.wide + (size_t)cur_a + lane;
                double bb[6];
                for (int c = 0; c < 6; c++) bb[c] = cr->b[c];
                const uint32_t rec_contained = cr->contained;
                ca = cr->a;
                cb = cr->m;
                const double d2c = boxDistance2(bb, p);
                push = d2c <= max_distance2;
                cd2 = floatBelow(d2c);
                // linear-octree.cpp:96-100; rounded UP to float: still an upper bound of the k-th distance
                if (push && rec_contained >= k) corner = floatAbove(boxMaxDistance2(bb, p));
            }
            const double best_corner = (double)waveMinPosF(corner);
            if (best_corner < max_distance2) max_distance2 = best_corner;
            // the bound of THIS step already applies to its records (the reference tightens while it loops over the children,
            // linear-octree.cpp:91-101); with up to 64 records per step it also keeps the frontier small
            push = push && (double)cd2 <= max_distance2;
            // keep the pushed children: own slots first
            if (push && f_b[0] == 0u) {
                f_d2[0] = cd2; f_a[0] = ca; f_b[0] = cb;
                push = false;
            } else if (push && f_b[1] == 0u) {
                f_d2[1] = cd2; f_a[1] = ca; f_b[1] = cb;
                push = false;
            }
            unsigned long long pmask = waveBallot(push);  // (rare) both slots of the tester taken: any free slot of the wave
            while (pmask) {
                const int src = __ffsll((long long)pmask) - 1;
                pmask &= pmask - 1;
                const float d = bitsFloat(waveRead(floatBits(cd2), src));
                const uint32_t a = waveRead(ca, src), b = waveRead(cb, src);
                const unsigned long long free0 = waveBallot(f_b[0] == 0u);
                if (free0) {
                    if ((int)lane == __ffsll((long long)free0) - 1) {
                        f_d2[0] = d; f_a[0] = a; f_b[0] = b;
                    }
                } else {
                    const unsigned long long free1 = waveBallot(f_b[1] == 0u);
                    if (free1) {
                        if ((int)lane == __ffsll((long long)free1) - 1) {
                            f_d2[1] = d; f_a[1] = a; f_b[1] = b;
                        }
                    } else if (kSpillRegs ? (W.spill && sp_n < kWaveSpill) : waveSpillPush(W, d, a, b)) {  // every register slot taken: to the list in memory
                        if constexpr (kSpillRegs) {
                            if (lane == 0) {
                                uint32_t* e = W.spill + 3u * sp_n;
                                __hip_atomic_store(e + 0, floatBits(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(e + 1, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(e + 2, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            sp_n++;
                        }
                        spilled = true;
                    } else {
                        overflow = kKnnOverflowFlag;
                    }
                }
            }
        }
        // pop the nearest octant of the frontier
        float mine = f_d2[0] < f_d2[1] ? f_d2[0] : f_d2[1];
        float best = waveMinPosF(mine);
        bool finished = false;
        while (!(best < INFINITY) || (double)best > max_distance2) {  // frontier empty, or nothing left within the bound (linear-octree.cpp:113)
            if (!kSpillRegs && !spilled) {  // (LDS form: one flag is all the search loop carries of the list)
                finished = true;
                break;
            }
            uint32_t* list = W.spill;
            if constexpr (!kSpillRegs) sp_n = waveSpillState(W, list);
            if (sp_n == 0u) {
                finished = true;
                break;
            }
            // what the registers hold is beyond the bound: drop it, and take the last (up to) 128 entries of the list instead
            __threadfence();
            const uint32_t take = sp_n < 128u ? sp_n : 128u;
            for (int s2 = 0; s2 < 2; s2++) {
                const uint32_t j = lane + 64u * (uint32_t)s2;
                f_d2[s2] = INFINITY;
                f_a[s2] = kNone;
                f_b[s2] = 0u;
                if (j < take) {
                    const uint32_t* e = list + 3u * (sp_n - 1u - j);
                    const float d = bitsFloat(__hip_atomic_load(e + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if ((double)d <= max_distance2) {
                        f_d2[s2] = d;
                        f_a[s2] = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        f_b[s2] = __hip_atomic_load(e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            sp_n -= take;
            if constexpr (!kSpillRegs) waveSpillSetCount(W, sp_n);
            mine = f_d2[0] < f_d2[1] ? f_d2[0] : f_d2[1];
            best = waveMinPosF(mine);
        }
        if (finished) break;
        const unsigned long long owner = waveBallot(mine == best);
        const int ol = __ffsll((long long)owner) - 1;
        const int which = f_d2[0] <= f_d2[1] ? 0 : 1;
        cur_a = waveRead(which == 0 ? f_a[0] : f_a[1], ol);
        cur_b = waveRead(which == 0 ? f_b[0] : f_b[1], ol);
        if ((int)lane == ol) {
            if (which == 0) { f_d2[0] = INFINITY; f_b[0] = 0u; }
            else { f_d2[1] = INFINITY; f_b[1] = 0u; }
        }
    }
    // exact selection, once: shrink to the entries that can still matter (k of them plus the few that share the k-th key's
    // leading bits), then drop the largest until k are left — instead of a 63-step search for the exact k-th key
    if (H.on && count > k) {
        const uint32_t n = histSelectK<R>(W, H, count, k, r2_max);
        if (n) return n;
    }
    if (count > k) {
        double bound;
        count = waveSelectBound<R>(W, count, k, bound, 2u);
    }
    if (count > 64u || count > k + 4u) return waveSelectK<R>(W, count, k, r2_max);  // a crowd at the k-th distance: the general selection
    return waveTrimToK(W, count, k, r2_max);
}

// ---- Estimate requests staged in memory ------------------------------------------------------------------------------------
// What Interaction::BSDF reads of the asking lane's Interaction (interaction.cpp:56-153), written by the lane to its own record
// before the estimates of a wave are served: position, out, the shading frame, n1, n2, R, T, type / inside, material. The
// serving loop reads the asking lane's record with wave-uniform loads AFTER the search and moves it to scalar registers. Round 2
// broadcast the fields out of the asking lane's vector registers (v_readlane): the whole Interaction then had to stay in VGPRs
// (or be reloaded from scratch) inside the serving loop, and the 128-VGPR instance ran ~90 scratch instructions per search
// (23 KB of spill traffic per search next to 13 KB of photons and octant records: measured 1.93 x the algorithmic bytes).
// With the record in memory nothing of the path state is touched between the first and the last search of a wave, so the
// register allocator spills it ONCE around the loop.
constexpr uint32_t kStageDoubles = 24;  // 19 values + flags + material (21), then the estimate coming back (3)
constexpr uint32_t kStageResult = 21;

template <bool L>
__device__ inline void stageInteraction(double* rec, const InteractionT<L>& ia) {
    double2* r = reinterpret_cast<double2*>(rec);
    r[0] = double2{ia.position.x, ia.position.y};
    r[1] = double2{ia.position.z, ia.out.x};
    r[2] = double2{ia.out.y, ia.out.z};
    r[3] = double2{ia.shading_cs.c0.x, ia.shading_cs.c0.y};
    r[4] = double2{ia.shading_cs.c0.z, ia.shading_cs.c1.x};
    r[5] = double2{ia.shading_cs.c1.y, ia.shading_cs.c1.z};
    r[6] = double2{ia.shading_cs.c2.x, ia.shading_cs.c2.y};
    r[7] = double2{ia.shading_cs.c2.z, ia.n1};
    r[8] = double2{ia.n2, ia.R};
    union { cptr<mcrt_material, L> p; unsigned long long u; } c;
    c.u = 0ull;
    c.p = ia.material;
    r[9] = double2{ia.T, __longlong_as_double((long long)(((unsigned long long)(uint32_t)ia.type) | (ia.inside ? 0x100000000ull : 0ull)))};
    r[10] = double2{__longlong_as_double((long long)c.u), 0.0};
}

// a value every lane holds (loaded from a wave-uniform address) moved to scalar registers
__device__ inline double uniformD(double v) {
    union { double d; int u[2]; } c;
    c.d = v;
    c.u[0] = __builtin_amdgcn_readfirstlane(c.u[0]);
    c.u[1] = __builtin_amdgcn_readfirstlane(c.u[1]);
    return c.d;
}

template <bool L>
__device__ inline void loadStagedInteraction(const double* rec, InteractionT<L>& q) {  // rec wave-uniform
    const double2* r = reinterpret_cast<const double2*>(rec);
    const double2 a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3], a4 = r[4], a5 = r[5], a6 = r[6], a7 = r[7], a8 = r[8], a9 = r[9], a10 = r[10];
    q.position = d3{uniformD(a0.x), uniformD(a0.y), uniformD(a1.x)};
    q.out = d3{uniformD(a1.y), uniformD(a2.x), uniformD(a2.y)};
    q.shading_cs.c0 = d3{uniformD(a3.x), uniformD(a3.y), uniformD(a4.x)};
    q.shading_cs.c1 = d3{uniformD(a4.y), uniformD(a5.x), uniformD(a5.y)};
    q.shading_cs.c2 = d3{uniformD(a6.x), uniformD(a6.y), uniformD(a7.x)};
    q.n1 = uniformD(a7.y);
    q.n2 = uniformD(a8.x);
    q.R = uniformD(a8.y);
    q.T = uniformD(a9.x);
    const unsigned long long fl = (unsigned long long)__double_as_longlong(uniformD(a9.y));
    q.type = (int)(uint32_t)fl;
    q.inside = (fl >> 32) != 0ull;
    union { cptr<mcrt_material, L> p; unsigned long long u; } c;
    c.u = (unsigned long long)__double_as_longlong(uniformD(a10.x));
    q.material = c.p;
}

// The sum of estimateGlobalRadiance / estimateCausticRadiance (photon-mapper.cpp:343-391) over the n photons in (d2, idx) for
// the Interaction q (wave-uniform): the photons are evaluated by n lanes in parallel, the contributions are summed by a wave
// reduction. r2 = photons.top().distance2, the farthest of the k. All lanes must call; returns the estimate in every lane.
template <bool L>
__device__ inline d3 waveEvalPhotons(const InteractionT<L>& q, const PhotonMapViewW& map, bool caustic, const MCRT_LDS_AS double* d2,
                                     const MCRT_LDS_AS uint32_t* idx, uint32_t n, double r2) {
    const uint32_t lane = __lane_id();
    MCRT_LOCKSTEP();  // (the result was compacted by other lanes)
    d3 sum = splat(0.0);
    const double inv_max_squared_radius = 1.0 / r2;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        d3 contrib = splat(0.0);
        if (j < n) {
            const float* ph = map.base.photons + (size_t)idx[j] * 8;
            d3 bsdf_absIdotN;
            double bsdf_pdf;
            if (interactionBSDF(q, bsdf_absIdotN, photonDirection(ph), bsdf_pdf)) {
                const d3 flux = d3{(double)ph[0], (double)ph[1], (double)ph[2]};
                if (caustic) {
                    const double wp = gmax(0.0, 1.0 - sqrt(d2[j] * inv_max_squared_radius));
                    contrib = (flux * bsdf_absIdotN * wp) / bsdf_pdf;
                } else {
                    contrib = flux * bsdf_absIdotN / bsdf_pdf;
                }
            }
        }
        sum = sum + d3{waveSumD(contrib.x), waveSumD(contrib.y), waveSumD(contrib.z)};
    }
    return caustic ? 3.0 * sum * inv_max_squared_radius * kInvPi : sum / (r2 * kPi);
}

// The radiance estimate for every lane of the wave that asks for one (`want`), served one query at a time by the whole
// wave from the staged records (`stage_wave` = the record of the wave's lane 0; every asking lane has called stageInteraction
// and the stores are visible: __threadfence_block() in between). Returns the estimate to the asking lane (zero elsewhere).
template <bool L, int R = kWaveRows>
__device__ inline d3 waveEstimate(bool want, double* stage_wave, const PhotonMapViewW& map, uint32_t k, bool caustic,
                                  const WaveKnnLds& W, uint32_t& searches, uint32_t& octant_visits, uint32_t& overflow) {
    const uint32_t lane = __lane_id();
    unsigned long long mask = waveBallot(want);
    if (!mask) return splat(0.0);
    if (lane == 0) searches += (uint32_t)__popcll(mask);
    // Nothing per-lane is carried through the serving loop: the estimate goes back through the asking lane's record (its last
    // three values), written by one lane, read by the asking lane after the loop.
    while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        double* rec = stage_wave + (size_t)src * kStageDoubles;
        // the asking lane's position now; the rest of its record only once the search is over (fewer wave-uniform registers
        // live across the search)
        const double2 p01 = reinterpret_cast<const double2*>(rec)[0];
        const d3 qpos = d3{uniformD(p01.x), uniformD(p01.y), uniformD(rec[2])};
        double r2 = 0.0;
        const uint32_t n = waveKnnSearch<R, L>(map, qpos, k, W, r2, overflow, octant_visits);  // (L: the scene is LDS-resident)
        d3 sum = splat(0.0);
        if (n) {  // else photons.empty(): the estimate is zero (photon-mapper.cpp:347, :374)
            InteractionT<L> q;
            loadStagedInteraction(rec, q);
            sum = waveEvalPhotons(q, map, caustic, W.d2, W.idx, n, r2);
        }
        if (lane == 0) {
            rec[kStageResult] = sum.x;
            rec[kStageResult + 1] = sum.y;
            rec[kStageResult + 2] = sum.z;
        }
    }
    __threadfence_block();
    d3 result = splat(0.0);
    if (want) {
        const double* mine = stage_wave + (size_t)lane * kStageDoubles + kStageResult;
        result = d3{mine[0], mine[1], mine[2]};
    }
    return result;
}

}  // namespace mcrt

#endif  // __HIPCC__
