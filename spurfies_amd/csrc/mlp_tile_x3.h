// Device helpers of the bf16-piece ("x3") tile engine shared by the geometry and colour kernels: fp32-exact products from three
// bf16 pieces per operand on v_mfma_f32_32x32x16_bf16 (see geo_mlp.hip for the arithmetic argument).
// Activations live in LDS as three bf16 planes [piece][row][k] (row stride X3_LDP); weights stream from L2 as piece fragments
// [wave][k16][m][piece][lane] x 8 bf16.  Transposed product D[feature][row] += W[feature][k] X[k][row].
#pragma once
#include "mlp_tile.h"

namespace spf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef const __attribute__((address_space(1))) bf16x8* gx3;

constexpr int X3_LDP = 264;                       // plane row stride in bf16 (528 B: 16-B aligned, off the 256-B bank period)
constexpr int X3_PLANE = 64 * X3_LDP;

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)x;                                // round to nearest even (v_cvt_pk_bf16_f32)
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    c = (__bf16)(r1 - (float)b);
}


// acc[m][n] (features 64w + 32m.., rows 32n..) += W X over T k16-steps.  wp: this wave's fragments of the layer, + lane.
// `first`: the layer's k-step-0 fragments, requested by the previous layer's GEMM (its last k-step) so that their L2 round trip
// is not exposed behind the barriers; returns the k-step-0 fragments of `next_wp` (or `first`).
struct WFrag3 {
    bf16x8 w[2][3];
};
__device__ __forceinline__ WFrag3 load_wfrag3(gx3 wp) {
    WFrag3 f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[m][p] = wp[(m * 3 + p) * 64];
    return f;
}

template <int T>
__device__ __forceinline__ WFrag3 gemm_x3(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][2], const WFrag3& first, gx3 next_wp) {
    const int j = lane & 31, kg = lane >> 5;
    bf16x8 wa[2][3], wn[2][3];
    WFrag3 nxt = first;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) wa[m][p] = first.w[m][p];
#pragma unroll 2
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) wn[m][p] = wp[((t + 1) * 6 + m * 3 + p) * 64];
        } else if (next_wp) {
            nxt = load_wfrag3(next_wp);
        }
        bf16x8 xb[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[n][p] = *reinterpret_cast<const bf16x8*>(X + p * X3_PLANE + (32 * n + j) * X3_LDP + 16 * t + 8 * kg);
        // smallest terms first; four accumulators alternate
#define SPF_X3(PW, PX)                                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PW], xb[0][PX], acc[0][0], 0, 0, 0);                   \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PW], xb[1][PX], acc[0][1], 0, 0, 0);                   \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PW], xb[0][PX], acc[1][0], 0, 0, 0);                   \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PW], xb[1][PX], acc[1][1], 0, 0, 0);
        SPF_X3(2, 0) SPF_X3(0, 2) SPF_X3(1, 1) SPF_X3(1, 0) SPF_X3(0, 1) SPF_X3(0, 0)
#undef SPF_X3
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) wa[m][p] = wn[m][p];
    }
    return nxt;
}

// write 4 consecutive features of one row as three bf16 quads
__device__ __forceinline__ void store_quad_x3(__bf16* X, int row, int f0, const float (&v)[4]) {
    bf16x4 pa, pb, pc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 a, b, c;
        split3(v[e], a, b, c);
        pa[e] = a; pb[e] = b; pc[e] = c;
    }
    __bf16* dst = X + row * X3_LDP + f0;
    *reinterpret_cast<bf16x4*>(dst) = pa;
    *reinterpret_cast<bf16x4*>(dst + X3_PLANE) = pb;
    *reinterpret_cast<bf16x4*>(dst + 2 * X3_PLANE) = pc;
}


// [64][8 * NG] tile of the planes (exactly p1 + p2 + p3 per element) -> fp32 rows in HBM, coalesced (32 B per thread, a row's
// threads are consecutive): what the weight-gradient GEMM reads.
template <int NG>
__device__ __forceinline__ void store_tile_from_planes(const __bf16* X, float* __restrict__ dst, int ld_dst, int tid) {
    for (int idx = tid; idx < 64 * NG; idx += 256) {
        const int row = idx / NG, gc = idx % NG;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(X + row * X3_LDP + 8 * gc);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(X + X3_PLANE + row * X3_LDP + 8 * gc);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(X + 2 * X3_PLANE + row * X3_LDP + 8 * gc);
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = ((float)a[e] + (float)b[e]) + (float)c[e];
            hi[e] = ((float)a[e + 4] + (float)b[e + 4]) + (float)c[e + 4];
        }
        float* d = dst + (size_t)row * ld_dst + 8 * gc;
        *reinterpret_cast<f32x4*>(d) = lo;
        *reinterpret_cast<f32x4*>(d + 4) = hi;
    }
}

// one element (row, col) of the planes
__device__ __forceinline__ void store_one_x3(__bf16* X, int row, int col, float v) {
    __bf16 a, b, c;
    split3(v, a, b, c);
    X[row * X3_LDP + col] = a;
    X[X3_PLANE + row * X3_LDP + col] = b;
    X[2 * X3_PLANE + row * X3_LDP + col] = c;
}

}  // namespace spf
