// The link between mcrt_hip.hip and mcrt_hip_lean.hip (not part of the C ABI): the default path's kernels compiled a second time
// WITHOUT the material features most scenes do not use (MCRT_MAT_FEATURES_OFF, csrc/mcrt_shade.hpp). Global scope on purpose: the
// second translation unit renames the library's namespace, this header must read the same in both.
#pragma once

// The instances mcrt_hip_lean.hip holds; mcrt_lean_kernel(id) returns the address hipLaunchKernel takes (the host stub), or null.
enum McrtLeanKernelId {
    MCRT_LEAN_FLATK_512 = 0,   // renderKernelFlatK<512>
    MCRT_LEAN_FLATK_768,       // renderKernelFlatK<768>
    MCRT_LEAN_FLAT_512,        // renderKernel<path tracer, false, true, false, 1>  (flat scene whose cull records do not fit the argument block)
    MCRT_LEAN_FLAT_768,        // renderKernel<path tracer, false, true, false, 2>
    MCRT_LEAN_PM_1024_ALL,     // renderKernelPM<false, true, 1024>
    MCRT_LEAN_PM_512_ALL,      // renderKernelPM<false, true>
    MCRT_LEAN_SM,              // renderKernelSM<false, false>
    MCRT_LEAN_SM_ALL,          // renderKernelSM<false, true>
    MCRT_LEAN_SHADE,           // wfShadeKernel<false>
    MCRT_LEAN_SHADE_PM,        // wfShadeKernel<true>
    MCRT_LEAN_EMIT,            // emitKernel<false>
    MCRT_LEAN_EMIT_ALL,        // emitKernel<true>
    MCRT_LEAN_KNN_EVAL,        // wfKnnKernel<true>  (the pipeline's kNN launch that evaluates the estimates: Interaction::BSDF per photon)
    MCRT_LEAN_COUNT
};
// the material flag bits those instances are compiled without (include/mcrt.h: MCRT_MAT_ROUGH | MCRT_MAT_ROUGH_SPECULAR | MCRT_MAT_COMPLEX_IOR)
#define MCRT_LEAN_FEATURES_OFF ((1u << 0) | (1u << 1) | (1u << 6))
extern "C" const void* mcrt_lean_kernel(int id);
