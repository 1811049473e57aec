// Force-included (-include) into every translation unit of the REFERENCE when it is compiled,
// unmodified and in place, into oracle/_ref/ (see oracle/Makefile). Test infrastructure only.
//
// The reference seeds Sampler::global_seed (sampling/sampler.hpp:58) and Random::engine
// (sampling/sampling.hpp:50) from std::random_device, which makes renders non-reproducible.
// <random> is included here first, then the identifier is re-pointed at a deterministic
// stand-in, so both initialisers become constants without touching the reference sources.
// Seed: environment variable MCRT_REF_SEED (decimal or 0x-hex), default 0x12345678.
#pragma once
#include <cstdlib>
#include <random>

namespace std {
struct mcrt_fixed_random_device {
    using result_type = unsigned int;
    result_type operator()() const {
        const char* s = std::getenv("MCRT_REF_SEED");
        return s ? static_cast<result_type>(std::strtoul(s, nullptr, 0)) : 0x12345678u;
    }
    static constexpr result_type min() { return 0u; }
    static constexpr result_type max() { return 0xFFFFFFFFu; }
};
}  // namespace std
#define random_device mcrt_fixed_random_device
