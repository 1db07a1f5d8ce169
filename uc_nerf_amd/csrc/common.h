// Shared host-side helpers for libucnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ucnerf_hip.h"

namespace ucnerf {

char* last_error_buf();   // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(UCNERF_EHIP, "%s: %s", what, hipGetErrorString(e));
    return UCNERF_OK;
}

#define UCNERF_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::ucnerf::fail(UCNERF_EINVAL, __VA_ARGS__); } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

struct Mat34 { float m[12]; };
struct Mat33 { float m[9]; };

int device_cus();   // cached

}  // namespace ucnerf
