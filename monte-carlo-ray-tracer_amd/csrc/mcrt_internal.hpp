// Internal links between the translation units of libmcrt_hip.so (not part of the C ABI).
#pragma once

#include <string>

#include "../../include/mcrt.h"

struct mcrt_bvh;

namespace mcrt {
int ctxDevice(const mcrt_ctx* ctx);
void* ctxStream(const mcrt_ctx* ctx);  // the context's hipStream_t
int ctxFail(mcrt_ctx* ctx, int code, const std::string& msg);  // records the message for mcrt_last_error, returns code
// Run-time options of a context (mcrt_set_option; the MCRT_* environment variables only seed them at mcrt_create).
const char* ctxOpt(const mcrt_ctx* ctx, const char* key);       // value or nullptr when unset
long ctxOptL(const mcrt_ctx* ctx, const char* key, long dflt);  // integer value or dflt
bool ctxOptOn(const mcrt_ctx* ctx, const char* key);            // set and not 0
// mcrt_octree_gpu.hip: the octree BVH of `scene` with the per-surface work on the GPU of ctx (mcrt_bvh_shared.hpp)
int bvhOctreeGpu(mcrt_ctx* ctx, const mcrt_scene_desc* scene, struct ::mcrt_bvh* out);
// mcrt_sah_gpu.hip: the binned-SAH hierarchies level by level with the per-surface passes on the GPU of ctx (mcrt_sah_shared.hpp)
int bvhSahGpu(mcrt_ctx* ctx, const mcrt_scene_desc* scene, int arity, int bins, struct ::mcrt_bvh* out);
// mcrt_output.hip: refPow (mcrt_libm_pow.hpp, the output stage's one libm call) on device arrays, launched on `stream`: out[i] =
// pow(a[i], b[i]). The known-answer kernel behind mcrt_libm's MCRT_LIBM_POW; returns the hipError_t of the launch as an int.
int launchPowKat(void* stream, uint64_t n, const double* a, const double* b, double* out);
}  // namespace mcrt
