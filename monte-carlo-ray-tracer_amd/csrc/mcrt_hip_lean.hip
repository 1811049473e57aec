// libmcrt_hip.so, second kernel translation unit: the default path's kernels compiled WITHOUT the rough-diffuse (Oren-Nayar),
// rough-specular (GGX) and conductor-Fresnel branches (MCRT_MAT_FEATURES_OFF, csrc/mcrt_shade.hpp) - for scenes none of whose materials
// carries one of those flags (hexagon_room, water_caustics, ...: launchRender decides, csrc/mcrt_hip.hip). Same source, same arithmetic on
// every path such a scene can take - the frames are the full kernels' bits (tests/test_gpu_lean_kernels.py) - but the register peak of
// the shading block is gone: the flat megakernel's 512-lane form spills nothing instead of 13 registers, the photon-mapping kernels ~100
// fewer, and a frame of hexagon_room renders 3.4 %, of its photon-mapped variant 7 %, of water_caustics 2 % faster
// (profiles/r06_ab_feature_strip*.log). No host code here but the table of kernel addresses.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mcrt_lean.hpp"
#define MCRT_MAT_FEATURES_OFF MCRT_LEAN_FEATURES_OFF
// the lean kNN launch's occupancy: with 17 spilled registers at 96 VGPRs it has room for a sixth wave per SIMD (80 VGPRs, 24 spilled) - C5
// through the pipeline 777 -> 739 ms, a seventh changes nothing (profiles/r06_ab_lean_knn_occupancy.log); the full instance stays at 5
#define MCRT_KNN_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
// Every inline function of the headers below exists in mcrt_hip.hip's objects too, compiled with all features: this unit's copies live
// in a namespace of their own so that the linker never folds the two (the C ABI's identifiers, mcrt_*, are not touched by the macro).
#define mcrt mcrt_lean

#include "../../include/mcrt.h"
#include "mcrt_integrator.hpp"
#include "mcrt_lanesm.hpp"
#include "mcrt_qbvh.hpp"
#include "mcrt_wavefront.hpp"
#include "mcrt_waveknn.hpp"
#include "mcrt_groupknn.hpp"
#include "mcrt_widerec.hpp"

using namespace mcrt;

static_assert(MCRT_LEAN_FEATURES_OFF == (MCRT_MAT_ROUGH | MCRT_MAT_ROUGH_SPECULAR | MCRT_MAT_COMPLEX_IOR), "mcrt_lean.hpp names the bits by value");
static_assert(kMatFeaturesOff == MCRT_LEAN_FEATURES_OFF, "this unit compiles the three features out");

namespace {
namespace lean {  // (named: the kernels' symbols must differ from mcrt_hip.hip's - tools/device_code_hashes.py and the spill table list kernels by name)
#include "mcrt_kernels.hpp"
}  // namespace lean
}  // namespace

extern "C" const void* mcrt_lean_kernel(int id) {
    using namespace lean;
    constexpr int PT = MCRT_INTEGRATOR_PATH_TRACER;
    switch (id) {
        case MCRT_LEAN_FLATK_512: return reinterpret_cast<const void*>(renderKernelFlatK<>);
        case MCRT_LEAN_FLATK_768: return reinterpret_cast<const void*>(renderKernelFlatK<768>);
        case MCRT_LEAN_FLAT_512: return reinterpret_cast<const void*>(renderKernel<PT, false, true, false, 1>);
        case MCRT_LEAN_FLAT_768: return reinterpret_cast<const void*>(renderKernel<PT, false, true, false, 2>);
        case MCRT_LEAN_PM_1024_ALL: return reinterpret_cast<const void*>(renderKernelPM<false, true, 1024>);
        case MCRT_LEAN_PM_512_ALL: return reinterpret_cast<const void*>(renderKernelPM<false, true>);
        case MCRT_LEAN_SM: return reinterpret_cast<const void*>(renderKernelSM<false, false>);
        case MCRT_LEAN_SM_ALL: return reinterpret_cast<const void*>(renderKernelSM<false, true>);
        case MCRT_LEAN_SHADE: return reinterpret_cast<const void*>(wfShadeKernel<false>);
        case MCRT_LEAN_SHADE_PM: return reinterpret_cast<const void*>(wfShadeKernel<true>);
        case MCRT_LEAN_EMIT: return reinterpret_cast<const void*>(emitKernel<false>);
        case MCRT_LEAN_EMIT_ALL: return reinterpret_cast<const void*>(emitKernel<true>);
        case MCRT_LEAN_KNN_EVAL: return reinterpret_cast<const void*>(wfKnnKernel<true>);
        default: return nullptr;
    }
}
