// Host-side helpers of the C ABI: error text, argument checks, HIP call checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tchgeo.h"

namespace tg {

char *last_error_buffer();

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

} // namespace tg

#define TG_REQUIRE(cond, ...)                                                                                          \
    do {                                                                                                               \
        if (!(cond)) return tg::fail(TG_ERR_INVALID, __VA_ARGS__);                                                     \
    } while (0)

#define TG_HIP(call)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return tg::fail(TG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));              \
    } while (0)

#define TG_LAUNCH_CHECK()                                                                                              \
    do {                                                                                                               \
        hipError_t e_ = hipGetLastError();                                                                             \
        if (e_ != hipSuccess) return tg::fail(TG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_));          \
    } while (0)
