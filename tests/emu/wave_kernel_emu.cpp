// wfTraceKernel itself (csrc/mcrt_kernels.hpp: the persistent-wave trace kernel of the wavefront pipeline and of mcrt_intersect -
// queue dealt in blocks to the workgroups, LDS cursor, batched refills and hit stores, the gates, the top of the tree and the root
// staged in LDS, per-lane stacks in LDS + spill) run on the HOST: workgroups of several emulated wavefronts (wave_emu.hpp:
// cross-lane operations per wave, __syncthreads per workgroup), one workgroup after the other, the device source unchanged.
// Arguments are filled the way planTrace (mcrt_hip.hip) fills them. Test harness only.
#define MCRT_WAVE_EMU 1
#include "wave_emu.hpp"

#include "mcrt_emu.cpp"

#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_wbvh.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_groupknn.hpp"
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_widerec.hpp"

namespace {
alignas(64) unsigned char lds[160 * 1024];  // what `extern __shared__ unsigned char lds[]` of the kernels refers to here
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_kernels.hpp"

template <int kForm>
void launchTrace(const WfTraceArgs& a, const ArrayRays& rays, uint32_t grid, uint32_t waves) {
    for (uint32_t g = 0; g < grid; g++) {
        wemu::launch().block_idx = g;
        wemu::launch().block_dim = waves * 64u;
        wemu::launch().grid_dim = grid;
        wemu::runGroup((int)waves, [&](int) { wfTraceKernel<ArrayRays, true, kForm>(a, rays); });
    }
}
}  // namespace

extern "C" {

// Closest hits of n rays through wfTraceKernel<ArrayRays, count, form> (form: 0 first walk, 2 deferred leaves, 3 shared leaf step - the
// default; 1 = eight-wide nodes when the layout has them). grid workgroups of `waves` wavefronts; lds_blocks / lds_stack / refill /
// leaf_lanes / deal_shift as planTrace's options (lds_blocks 0xFFFFFFFF = as many as the tree has, up to 512). stats: the kernel's
// counters [kStatsWords] (rays at [1], node / primitive tests at [2] / [3], overflow at [5]). Returns 0, -100 on stack overflow.
int wemu_trace_kernel(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int form, uint32_t grid, uint32_t waves,
                      uint32_t lds_blocks, int lds_stack, int refill_lanes, int leaf_lanes, uint32_t deal_shift, double* out_t, uint32_t* out_surface,
                      double* out_uv, unsigned long long* stats_out) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    if (scene->num_nodes == 0 || grid == 0 || waves == 0 || waves > 16) return -200;
    if (form == 1 && E.L.wnodes.empty()) return -201;
    const uint32_t block = waves * 64u;
    const uint32_t depth = (uint32_t)std::max<int>(kMaxStackDepth, (int)E.L.stack_bound + 1);
    std::vector<SmStackEntry> spill((size_t)grid * block * depth);
    std::vector<unsigned long long> stats(kStatsWords + 32, 0ull);
    unsigned long long count = n, pop = 0;
    WfTraceArgs a;
    memset(&a, 0, sizeof(a));
    a.count = &count;
    a.pop = &pop;
    a.stats = stats.data();
    a.nodes = E.L.nodes64.data();
    a.qblocks = E.L.qblocks.data();
    a.wnodes = E.L.wnodes.empty() ? nullptr : E.L.wnodes.data();
    a.num_nodes = (uint32_t)E.L.nodes64.size();
    a.lds_blocks = form == 1 ? 0u : std::min<uint32_t>(lds_blocks, std::min<uint32_t>((uint32_t)E.L.qblocks.size(), 512u));
    a.q_root_a = E.L.q_root_a;
    a.q_root_m = E.L.q_root_m;
    a.prim = E.L.prim.data();
    a.spill = spill.data();
    a.total_lanes = grid * block;
    a.refill_lanes = refill_lanes;
    a.leaf_lanes = leaf_lanes;
    a.leaf_items = 1 << 20;
    a.min_inner = 8;
    a.lds_stack = lds_stack;
    a.max_stack = depth;
    a.deal_shift = deal_shift;
    const size_t lds_need = (size_t)a.lds_blocks * 64u + (size_t)lds_stack * block * sizeof(SmStackEntry) + 64u + waves * kShareMapBytes + 64u;
    if (lds_need > sizeof(lds)) return -202;
    ArrayRays rays;
    rays.start = start;
    rays.direction = direction;
    rays.out_t = out_t;
    rays.out_surface = out_surface;
    rays.out_uv = out_uv;
    if (form == 0) launchTrace<0>(a, rays, grid, waves);
    else if (form == 1) launchTrace<1>(a, rays, grid, waves);
    else if (form == 2) launchTrace<2>(a, rays, grid, waves);
    else launchTrace<3>(a, rays, grid, waves);
    if (stats_out) memcpy(stats_out, stats.data(), kStatsWords * sizeof(unsigned long long));
    return stats[5] ? -100 : 0;
}

}  // extern "C"
