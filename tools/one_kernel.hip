#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include "../include/mcrt.h"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_integrator.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_lanesm.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_qbvh.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_wavefront.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_waveknn.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_groupknn.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_widerec.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_layout.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_internal.hpp"
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_plan.hpp"
using namespace mcrt;
namespace {
#include "../monte-carlo-ray-tracer_amd/csrc/mcrt_kernels.hpp"
}
#ifndef ONE_KERNEL
#define ONE_KERNEL renderKernelFlatK<768>
#endif
void* one_kernel_addr() { return (void*)(ONE_KERNEL); }
