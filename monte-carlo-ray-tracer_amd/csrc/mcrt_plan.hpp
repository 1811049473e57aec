// Work-unit planning shared by the integrator launches (mcrt_hip.hip): how a frame is cut into passes of rows that fit
// the per-sample store, and a pixel's samples into chunks (the work units of RenderParams / WfFrame). Plain host C++,
// also built into tests/emu so that the rules are checked without a GPU.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace mcrt {

constexpr uint32_t kSampleBytes = 24;  // one sample's radiance in the store: 3 doubles
constexpr uint32_t kMinChunk = 4;      // a work unit keeps at least this many samples (unless the pixel has fewer)

struct PassPlan {
    uint32_t pass_rows;    // local rows per pass: a multiple of 8 (pixels are handed out in 8x8 tiles), at least 8
    uint64_t store_bytes;  // size of the store one pass needs
};

// As many rows per pass as a store of store_gb * 1e9 bytes holds.
inline PassPlan planPasses(uint32_t width, uint32_t owned_rows, uint32_t spp, double store_gb) {
    const uint64_t row_bytes = (uint64_t)width * spp * kSampleBytes;
    uint64_t rows = (uint64_t)(store_gb * 1e9) / row_bytes / 8 * 8;
    rows = std::max<uint64_t>(8, std::min<uint64_t>(rows, ((uint64_t)owned_rows + 7) / 8 * 8));
    PassPlan p;
    p.pass_rows = (uint32_t)rows;
    p.store_bytes = std::min<uint64_t>(rows, owned_rows) * row_bytes;
    return p;
}

struct ChunkPlan {
    uint32_t shift;  // units per pixel = 1 << shift
    uint32_t chunk;  // samples per unit = ceil(spp / units per pixel); the last unit(s) of a pixel may be short or empty
};

// The smallest power-of-two number of units per pixel that reaches `want`, as long as chunks keep `min_chunk` samples.
inline ChunkPlan planChunks(uint32_t spp, uint64_t want, uint32_t min_chunk = kMinChunk) {
    ChunkPlan c;
    c.shift = 0;
    while ((1ull << c.shift) < want && (spp >> (c.shift + 1)) >= min_chunk) c.shift++;
    c.chunk = (spp + (1u << c.shift) - 1u) >> c.shift;
    return c;
}

// Work units of the megakernels (one launch, lanes pop units from a counter): every unit costs its pop, the sampler's start at the
// pixel and the decode of the tile - on the hexagon_room frame about as much as three path samples. Round 4, ms per 1080p @ 256 spp
// shard of 1 / 2 / 4 / 8 (what one rank renders) with 4 / 16 / 64 units per pixel: 446.9 / 227.1 / 116.4 / 61.5, 445.3 / 225.4 /
// 113.2 / 57.2, - / 264.5 / 132.5 / 66.5. So: 128 units per lane for balance, but chunks of at least 16 samples; shorter chunks (down
// to kMinChunk) only when that leaves a lane fewer than 4 units (small frames).
inline ChunkPlan planChunksMega(uint32_t spp, uint64_t lanes, uint64_t pass_pixels, const char* chunks_override = nullptr) {
    if (chunks_override) return planChunks(spp, strtoull(chunks_override, nullptr, 0));  // option MCRT_CHUNKS: units per pixel
    const uint64_t pixels = std::max<uint64_t>(pass_pixels, 1);
    ChunkPlan c = planChunks(spp, (128 * lanes + pixels - 1) / pixels, 16);
    if ((pixels << c.shift) < 4 * lanes) c = planChunks(spp, (4 * lanes + pixels - 1) / pixels);
    return c;
}

// Units per pixel that give each of `consumers` (resident lanes, pool slots) `per_consumer` units; the option MCRT_CHUNKS
// (`chunks_override`, its value or null) overrides.
inline uint64_t unitsWanted(uint64_t consumers, uint64_t per_consumer, uint64_t pass_pixels, const char* chunks_override = nullptr) {
    if (chunks_override) return strtoull(chunks_override, nullptr, 0);
    return (per_consumer * consumers + pass_pixels - 1) / pass_pixels;
}

// Pool slots of the wavefront pipeline for a pass of `pass_paths` path samples, at most `max_slots` (option / memory), in multiples
// of `block` (the shade kernel's workgroup). A slot works through its path samples one after the other, so a pass takes about
// (samples per slot) x (bounces per path) shade + trace launches plus the tail of the longest paths; every launch lasts as long as its
// longest ray and, in the tail, pays a lane per slot whether it holds a path or not. Measured (round 4, ms per 1080p frame, samples per
// slot 64 / 32 / 16 / 8 / 4): spaceship 8 M paths 46 / 38 / 35 / 36 / 35, 33 M 85 / 78 / 77 / 85 / 90, 133 M 228 / 229 / 243 / 286 /
// 288; C3 8 M 77 / 53 / 42 / 38 / 36, 33 M 141 / 119 / 111 / 113 / 116, 133 M 390 / 371 / 378 / 410 / 411 - i.e. 2-4 M slots until
// the frame is large enough for more: paths / per_slot (48), at least floor_slots (2.5 M), never fewer than 4 samples per slot.
// per_slot_alone: the option MCRT_WF_SLOT_PATHS was given - it alone decides (A/B runs).
inline uint64_t planPoolSlots(uint64_t pass_paths, uint64_t max_slots, uint64_t block, uint64_t per_slot = 48, uint64_t floor_slots = 2500000,
                              bool per_slot_alone = false) {
    pass_paths = std::max<uint64_t>(pass_paths, 1);
    per_slot = std::max<uint64_t>(per_slot, 1);
    uint64_t want = std::max<uint64_t>(pass_paths / per_slot, per_slot_alone ? 1 : std::max<uint64_t>(floor_slots, 1));
    if (!per_slot_alone) want = std::min<uint64_t>(want, std::max<uint64_t>(pass_paths / 4, 1));
    const uint64_t slots = std::min<uint64_t>(max_slots, (want + block - 1) / block * block);
    return std::max<uint64_t>(slots, block);
}

// option MCRT_SAMPLE_STORE_GB, default 64 of the GPU's 288 GB: a 1080p @ 1024 spp frame (51 GB) is one pass (C3: 6.04 -> 5.94 s against
// four passes through 16 GB: every pass ends with a tail in which the pool runs dry); 4K @ 1024 spp is 204 GB, four passes
inline double sampleStoreGb(const char* option = nullptr) {
    return option ? atof(option) : 64.0;
}

}  // namespace mcrt
