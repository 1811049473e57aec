// Hash-based Owen-scrambled Sobol sampler of the reference (sampling/sampler.hpp:13-90,
// sampling/sobol.hpp:7-71) as a per-lane value type. Pure uint32 arithmetic; must match bit for bit.
//
// MI355X-specific restructuring: the reference evaluates a Sobol dimension with a 32-iteration
// loop XOR-ing one direction number per set index bit (sobol.hpp:65-69). The map index -> x is
// linear over GF(2), so it is tabulated per index BYTE: x = T[d][0][b0] ^ T[d][1][b1] ^ T[d][2][b2]
// ^ T[d][3][b3] with 6 dims x 4 x 256 u32 = 24 KiB, staged once per workgroup in LDS. Four LDS
// reads replace 32 dependent VALU iterations per dimension; the result is identical.
#pragma once

#include "mcrt_math.hpp"

namespace mcrt {

using SobolTab = MCRT_LDS_AS const uint32_t*;

constexpr int kSobolDims = 6;                       // dims 1..6 (dim 0 is the index itself)
constexpr int kSobolTableWords = kSobolDims * 4 * 256;  // 6144 u32 = 24 KiB

MCRT_HD uint32_t reverseBits(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

MCRT_HD uint32_t hash32(uint32_t x) {  // sampler.hpp:76-84 (hash-prospector 2-round)
    x ^= x >> 15;
    x *= 0xd168aaadu;
    x ^= x >> 15;
    x *= 0xaf723597u;
    x ^= x >> 15;
    return x;
}
MCRT_HD uint32_t hashCombine(uint32_t seed, uint32_t v) {  // sampler.hpp:87-90
    return seed ^ (v + 0x9e3779b9u + (seed << 6) + (seed >> 2));
}
MCRT_HD uint32_t owenScramble(uint32_t bit_reversed_x, uint32_t seed) {  // sampler.hpp:61-72
    bit_reversed_x ^= bit_reversed_x * 0x3d20adeau;
    bit_reversed_x += seed;
    bit_reversed_x *= (seed >> 16) | 1u;
    bit_reversed_x ^= bit_reversed_x * 0x05526c56u;
    bit_reversed_x ^= bit_reversed_x * 0x53a22864u;
    return reverseBits(bit_reversed_x);
}

// Host: direction numbers (sobol.hpp:18-54; Joe-Kuo new-joe-kuo-6.21201 dims 2..7, bit-reversed)
// folded into byte tables. out[kSobolTableWords].
inline void buildSobolByteTables(uint32_t* out) {
    static const uint32_t s[6] = {1, 2, 3, 3, 4, 4};
    static const uint32_t a[6] = {0, 1, 1, 2, 1, 4};
    static const uint32_t m[6][4] = {{1, 0, 0, 0}, {1, 3, 0, 0}, {1, 3, 1, 0}, {1, 1, 1, 0}, {1, 1, 3, 3}, {1, 3, 5, 13}};
    for (int dim = 0; dim < kSobolDims; dim++) {
        uint32_t V[32];
        for (uint32_t bit = 0; bit < s[dim]; bit++) V[bit] = m[dim][bit] << (31 - bit);
        for (uint32_t bit = s[dim]; bit < 32; bit++) {
            V[bit] = V[bit - s[dim]] ^ (V[bit - s[dim]] >> s[dim]);
            for (uint32_t k = 1; k < s[dim]; k++) V[bit] ^= (((a[dim] >> (s[dim] - 1 - k)) & 1u) * V[bit - k]);
        }
        for (uint32_t bit = 0; bit < 32; bit++) V[bit] = reverseBits(V[bit]);
        for (int byte = 0; byte < 4; byte++)
            for (uint32_t v = 0; v < 256; v++) {
                uint32_t x = 0;
                for (int b = 0; b < 8; b++)
                    if (v & (1u << b)) x ^= V[byte * 8 + b];
                out[(dim * 4 + byte) * 256 + v] = x;
            }
    }
}

struct Sampler {
    uint32_t base_seed, seed, sequence, bit_reversed_index, shuffled_index;

    MCRT_HD void initiate(uint32_t global_seed, uint32_t start_seed) {  // sampler.hpp:32-35
        base_seed = hashCombine(global_seed, hash32(start_seed));
    }
    MCRT_HD void setIndex(uint32_t index) {  // sampler.hpp:38-44
        sequence = 0u;
        seed = base_seed;
        bit_reversed_index = reverseBits(index);
        shuffled_index = index;
    }
    MCRT_HD void shuffle() {  // sampler.hpp:48-52
        seed = hashCombine(base_seed, hash32(++sequence));
        shuffled_index = owenScramble(bit_reversed_index, seed);
    }
    // The state after initiate(global_seed, start_seed), setIndex(index) and `shuffles` calls of shuffle(): every word is a
    // function of those four numbers, so a parked path keeps them instead of the sampler (mcrt_wavefront.hpp).
    MCRT_HD void restore(uint32_t global_seed, uint32_t start_seed, uint32_t index, uint32_t shuffles) {
        initiate(global_seed, start_seed);
        setIndex(index);
        if (shuffles != 0u) {
            sequence = shuffles - 1u;
            shuffle();
        }
    }
    // get<DIM>() (sampler.hpp:20-30). `tab` = byte tables (LDS on the GPU).
    MCRT_HD double get(int dim, MCRT_LDS_AS const uint32_t* tab) const {
        uint32_t x = shuffled_index;
        if (dim != 0) {
            MCRT_LDS_AS const uint32_t* t = tab + (dim - 1) * 1024;
            uint32_t i = shuffled_index;
            x = t[i & 255u] ^ t[256 + ((i >> 8) & 255u)] ^ t[512 + ((i >> 16) & 255u)] ^ t[768 + (i >> 24)];
        }
        return owenScramble(x, hashCombine(seed, hash32((uint32_t)dim))) * 0x1p-32;
    }
};

enum : int {  // sampling/sampling.hpp:59-76
    kDimPixel = 0, kDimLens = 2, kDimLight = 0, kDimBsdf = 3, kDimInteraction = 5, kDimAbsorb = 6
};

}  // namespace mcrt
