// The wave-cooperative photon search of the product (csrc/mcrt_waveknn.hpp: waveKnnSearch, the selection functions, the frontier
// spill list) run on the HOST, one emulated wavefront of 64 fibers per query (wave_emu.hpp) - test harness only, like mcrt_emu.cpp.
// The device code is included unchanged; the record lists are the ones mcrt_upload_photons builds (csrc/mcrt_widerec.hpp).
#define MCRT_WAVE_EMU 1
#include "wave_emu.hpp"

#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_widerec.hpp"

using namespace mcrt;

namespace {

template <int R, bool kRegs>
void searchOne(const PhotonMapViewW& map, d3 p, uint32_t k, uint32_t* spill, uint32_t* out_count, uint32_t* out_index, double* out_d2,
               uint32_t* overflow_out, unsigned long long* collectives) {
    std::vector<double> d2(waveCand(R));
    std::vector<uint32_t> idx(waveCand(R)), hist(kWaveHist), state(4, 0u);
    uint32_t overflow_any = 0;
    wemu::run([&](int lane) {
        WaveKnnLds W;
        W.d2 = d2.data();
        W.idx = idx.data();
        W.hist = hist.data();
        W.spill = kRegs ? spill : nullptr;
        if (!kRegs) {
            W.state = state.data();
            waveKnnInit(W, spill);
        }
        double r2 = 0.0;
        uint32_t overflow = 0, visits = 0;
        const uint32_t c = waveKnnSearch<R, kRegs>(map, p, k, W, r2, overflow, visits);
        waveSortResult<(R <= 4 ? 2 : R - 4)>(W, c);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) *out_count = c;
        for (uint32_t j = (uint32_t)lane; j < k; j += 64) {
            out_index[j] = j < c ? W.idx[j] : 0xFFFFFFFFu;
            out_d2[j] = j < c ? W.d2[j] : INFINITY;
        }
        if (overflow) overflow_any = 1;
    });
    if (overflow_any) *overflow_out = 1;
    (void)collectives;
}

}  // namespace

extern "C" {

// n queries against the map `m` (the reference's linear octree as mcrt_photon_map_get / PhotonMap hands it out). upload_k: the k the
// record lists are built for (mcrt_upload_photons' k_nearest_photons); rows: 4 or 16 candidate rows per wave; spill_mode: 0 = no
// spill list, 1 = its state in registers, 2 = in LDS. out_index / out_d2: [n][k] ascending (distance2, index), padded like mcrt_knn.
// Returns 0, or 1 / 2 from buildWideRecords, 3 for a bad argument; *overflow = 1 if any search raised the frontier-overflow flag.
int wemu_knn(const mcrt_photon_map_desc* m, uint32_t upload_k, uint64_t n, const double* pts, uint32_t k, int rows, int spill_mode,
             uint32_t* out_count, uint32_t* out_index, double* out_d2, uint32_t* overflow) {
    if (!m || !pts || k == 0 || (rows != 4 && rows != 16) || k > waveMaxK(rows)) return 3;
    *overflow = 0;
    const size_t no = m->num_octants;
    std::vector<uint32_t> start(no), contained(no);
    for (size_t i = 0; i < no; i++) {
        start[i] = (uint32_t)m->octant_start_data[i];
        contained[i] = (uint32_t)m->octant_contained_data[i];
    }
    std::vector<WideRec> wide;
    uint32_t root_a = 0, root_m = 0;
    if (no)
        if (const int rc = buildWideRecords(m, contained.data(), upload_k ? upload_k : 1u, wide, root_a, root_m)) return rc;
    std::vector<PhotonPos> pos((size_t)m->num_photons);
    for (size_t i = 0; i < pos.size(); i++) pos[i] = PhotonPos{m->photons[8 * i + 3], m->photons[8 * i + 4], m->photons[8 * i + 5]};
    PhotonMapViewW map;
    memset(&map, 0, sizeof(map));
    map.base.num_octants = m->num_octants;
    map.base.num_photons = m->num_photons;
    map.base.octant_bounds = m->octant_bounds;
    map.base.octant_start = start.data();
    map.base.octant_contained = contained.data();
    map.base.octant_next = m->octant_next_sibling;
    map.base.octant_leaf = m->octant_leaf;
    map.base.photons = m->photons;
    map.wide = wide.data();
    map.root_a = root_a;
    map.root_m = root_m;
    map.pos = pos.data();
    std::vector<uint32_t> spill((size_t)3 * kWaveSpill);
    for (uint64_t q = 0; q < n; q++) {
        const d3 p = d3{pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]};
        uint32_t* sp = spill_mode ? spill.data() : nullptr;
        uint32_t *oc = out_count + q, *oi = out_index + q * k;
        double* od = out_d2 + q * k;
        if (rows == 4) {
            if (spill_mode == 2) searchOne<4, false>(map, p, k, sp, oc, oi, od, overflow, nullptr);
            else searchOne<4, true>(map, p, k, sp, oc, oi, od, overflow, nullptr);
        } else {
            if (spill_mode == 2) searchOne<16, false>(map, p, k, sp, oc, oi, od, overflow, nullptr);
            else searchOne<16, true>(map, p, k, sp, oc, oi, od, overflow, nullptr);
        }
    }
    return 0;
}

}  // extern "C"
