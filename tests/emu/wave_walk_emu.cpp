// The trace kernels' wave-synchronous tree walk with the shared leaf step (csrc/mcrt_sharedleaf.hpp: traceWalkShared - deferred leaves
// tested by the whole wave, one pop site, the stack's top cached in registers) run on the HOST: 64 rays per emulated wavefront
// (wave_emu.hpp), the device source unchanged. Scene set-up (layout, quantised blocks, stacks) is mcrt_emu.cpp's. Test harness only.
#define MCRT_WAVE_EMU 1
#include "wave_emu.hpp"

#include "mcrt_emu.cpp"

namespace {
inline uint32_t laneId() { return __lane_id(); }
#include "../../monte-carlo-ray-tracer_amd/csrc/mcrt_sharedleaf.hpp"
}  // namespace

extern "C" {

// Closest hits of n rays, 64 per wave; `holes`: every holes-th lane of a wave carries no ray (valid = false), 0 = none.
// which: 0 = traceWalkShared (shared leaf step), 1 = traceWalkQ (every lane its own leaf step). Returns 0, -100 on stack overflow.
int wemu_intersect(const mcrt_scene_desc* scene, uint64_t n, const double* start, const double* direction, int which, int holes,
                   double* out_t, uint32_t* out_surface, double* out_uv) {
    Emu E;
    if (int rc = setup(E, scene, 0)) return rc;
    if (scene->num_nodes == 0) return -200;
    QTrace qt;
    qt.init(E, scene);
    const int depth = qt.stk.max_depth;
    std::vector<SmStackEntry> s_lds((size_t)kLdsStackDepth * 64), s_spill((size_t)(depth - kLdsStackDepth) * 64);
    std::vector<uint8_t> map(kShareMapBytes);
    uint32_t overflow = 0;
    // a wave takes the rays a stride apart, so that holes do not drop rays: ray index = wave base + position among the valid lanes
    uint64_t next = 0;
    while (next < n) {
        uint64_t idx[64];
        bool valid[64];
        for (int l = 0; l < 64; l++) {
            valid[l] = !(holes && (l % holes) == 0) && next < n;
            idx[l] = valid[l] ? next++ : 0;
        }
        wemu::run([&](int lane) {
            SmStack stk = qt.stk;
            stk.lds = s_lds.data() + lane;
            stk.lds_stride = 64;
            stk.spill = s_spill.data() + lane;
            stk.spill_stride = 64;
            TraceCounters cnt = {0, 0, 0, 0};
            const uint64_t i = idx[lane];
            Ray ray = makeRay(ld3(start + 3 * i), ld3(direction + 3 * i), 1.0);
            Hit h;
            if (which == 0) {
                h = traceWalkShared<true>(qt.sv, qt.qv, stk, valid[lane], ray, false, nullptr, cnt, map.data());
            } else {
                hitInit(h, kDblMax);
                // (traceWalkQ has no `valid`: lanes without a ray trace a copy of ray 0 and drop the result)
                h = traceWalkQ<true>(qt.sv, qt.qv, stk, ray, false, nullptr, cnt);
            }
            if (valid[lane]) {
                out_t[i] = h.t;
                out_surface[i] = h.surface;
                out_uv[2 * i] = h.u;
                out_uv[2 * i + 1] = h.v;
            }
            if (cnt.overflow) overflow = 1;
        });
    }
    return overflow ? -100 : 0;
}

}  // extern "C"
