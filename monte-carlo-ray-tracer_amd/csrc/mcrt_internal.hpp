// Internal links between the translation units of libmcrt_hip.so (not part of the C ABI).
#pragma once

#include <string>

#include "../../include/mcrt.h"

namespace mcrt {
int ctxDevice(const mcrt_ctx* ctx);
int ctxFail(mcrt_ctx* ctx, int code, const std::string& msg);  // records the message for mcrt_last_error, returns code
}  // namespace mcrt
