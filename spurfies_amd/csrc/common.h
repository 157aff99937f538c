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

// Order-independent accumulation for the latent-gradient scatters (the `*_fixed` arguments of the C ABI): every fp32 term is added
// as a 2^-48 fixed-point integer (float -> double -> x 2^48 -> nearest int64: exact for terms whose last mantissa bit is >= 2^-48,
// i.e. |v| >= 2^-25; smaller terms round at 3.6e-15 absolute; range |sum| < 2^13) with 64-bit integer atomics.  Integer addition is
// associative, so the result does not depend on the order the atomics land in — run-to-run reproducible, unlike float atomics —
// and spf_fixed_accumulate rounds the sum to fp32 once (more accurate than any fp32 summation order).
// Non-finite (or absurd, |v| >= 2^14) terms are recorded OUT OF BAND: the int64 word immediately BEFORE the accumulator array
// (acc[-1], part of the buffer contract, include/spurfies_hip.h) is its status word; such a term sets it with an atomic OR and adds
// nothing, and spf_fixed_accumulate then turns the WHOLE destination into NaN — the optimiser's skip-on-non-finite guard sees it
// whatever the number or the signs of the offending terms (a wrapping in-band marker cancels: four +2^62 sum to 0).
constexpr double FIXED_SCALE = 281474976710656.0;              // 2^48
constexpr double FIXED_LIMIT = 4611686018427387904.0;          // 2^62
#ifdef __HIPCC__
__device__ __forceinline__ void fixed_add(long long* base, size_t i, float v) {
    const double d = (double)v * FIXED_SCALE;
    if (!(fabs(d) < FIXED_LIMIT)) {                            // NaN, Inf, |v| >= 2^14
        atomicOr(reinterpret_cast<unsigned long long*>(base - 1), 1ull);
        return;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(base + i), (unsigned long long)__double2ll_rn(d));
}
#endif

}  // namespace spf
