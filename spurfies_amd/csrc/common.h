// Shared host/device helpers for libspurfies_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/spurfies_hip.h"

namespace spf {

constexpr int WAVE = 64;

// thread-local error text behind spf_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define SPF_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) return spf::fail(SPF_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define SPF_LAUNCH_CHECK(name)                                                                  \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) return spf::fail(SPF_EHIP, "launch %s: %s", name, hipGetErrorString(e__)); \
    } while (0)

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace spf
