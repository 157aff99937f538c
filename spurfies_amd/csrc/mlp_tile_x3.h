// Device helpers of the bf16-piece ("x3") tile engine shared by the geometry and colour kernels: fp32-class products (<= 2 ulp per product) from three
// bf16 pieces per operand on v_mfma_f32_32x32x16_bf16 (see geo_mlp.hip for the arithmetic argument).
// Activations live in LDS as three bf16 planes [piece][row][k] (row stride X3_LDP); weights stream from L2 as piece fragments
// [wave][k16][m][piece][lane] x 8 bf16.  Transposed product D[feature][row] += W[feature][k] X[k][row].
#pragma once
#include "mlp_tile.h"

namespace spf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef const __attribute__((address_space(1))) bf16x8* gx3;
// ---- "H2" arithmetic (round 6): an fp32 operand as TWO fp16 pieces, x = h1 + 2^-11 h2 with h1 = fp16(x), h2 = fp16((x - h1) 2^11) (both round to
// nearest; the difference and the scaling are exact in fp32): 22 mantissa bits instead of fp32's 24.  A product takes THREE piece products — h1 g1
// into the main accumulator, h1 g2 + h2 g1 into a second one that is scaled by 2^-11 once per layer (every piece product is exact in the MFMA's fp32
// accumulation: 11 x 11 bits) — i.e. HALF the matrix instructions of the six-product bf16 scheme and two thirds of its operand bytes; the dropped
// h2 g2 is <= 2^-22 of the product.  Per product: |error| <= 3 x 2^-22 ~ 7e-7 relative (bf16 x 3: ~ 2e-7; a true fp32 product: 6e-8).  Range is
// fp16's: operands above 65504 overflow (no MLP activation of this model comes near), operands below 6e-5 keep an ABSOLUTE accuracy of 1.5e-11
// (fp16 subnormals are honoured by the matrix pipe).  The pieces live in the same registers / LDS planes / fragment slots as bf16 pieces 0 and 1.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float H2_EPS = 1.0f / 2048.0f;

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
    bf16x8 w[2][3];       // k-step 0
    bf16x8 w1[2][3];      // k-step 1
};
__device__ __forceinline__ WFrag3 load_wfrag3(gx3 wp) {
    WFrag3 f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            f.w[m][p] = wp[(m * 3 + p) * 64];
            f.w1[m][p] = wp[(6 + m * 3 + p) * 64];
        }
    return f;
}

// One k-step: [requests: weights of k-step t + 2 (LW) / the next layer's first fragments (LN), LDS operands of k-step t + 1 (LX)],
// then this k-step's 24 MFMAs.  Weight fragments are requested TWO k-steps ahead (a k-step is 768 cycles, less than a loaded L2
// round trip, and the wave has its SIMD to itself: nothing else covers a wait), LDS operands one k-step ahead.  Operands rotate
// through three register buffers each, addressed by the compile-time phase R = t mod 3: a rotation by register moves would have to
// wait for the youngest request.  The order is pinned: left alone, the scheduler sinks the requests to the end of the k-step.
template <int NT = 2>
struct X3Regs {
    bf16x8 w[3][2][3], x[3][NT][3];
};
// SWAP = false: transposed product, acc[m][n] = (features 32 m.., rows 32 n..) = W X; SWAP = true: acc[m][n] = (rows 32 m.., features
// 32 n..) = X^T W^T, the same fragments with the operand roles exchanged (a lane then owns a feature column strip).
// NT = number of 32-row tiles of the activation operand a wave multiplies: 2 = the 64-row tile (every weight fragment feeds two MFMAs), 1 = the
// HALF-HEIGHT 32-row tile of round 5 (one MFMA per fragment: the k-step is 12 MFMAs = 384 cycles for the same 6 KB of fragments, i.e. at the
// CU's L1 fill rate — twice the tiles at ~55 % of a tile's time each: for launches whose 64-row tiles would not fill the chip once).
// NPC = pieces per operand that take part: 3 = the fp32-class product (six piece products), 2 = the REDUCED product of round 6 (pieces 0 and 1
// of both operands, products (1,0) (0,1) (0,0): ~16 mantissa bits, half the MFMAs and two thirds of the operand traffic) — offered to the
// evaluation sampler's SDF-only passes only (include/spurfies_hip.h: SPF_ARITH_LITE), never to a pass whose values are rendered or differentiated.
template <int R, bool LW, bool LX, int LN, bool SWAP = false, int LDP = X3_LDP, bool FIRST = false, int NT = 2, int NPC = 3, bool H2 = false>
__device__ __forceinline__ void x3_step(const __bf16* xp, gx3 wp, int t, f32x16 (&acc)[2][NT], X3Regs<NT>& r, WFrag3& nxt, gx3 next_wp,
                                        f32x16 (&accc)[2][NT]) {
    static_assert(!H2 || NPC == 2, "H2: two fp16 pieces");
    constexpr int R1 = (R + 1) % 3, R2 = (R + 2) % 3;
    if (LW) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < NPC; ++p) r.w[R2][m][p] = wp[((t + 2) * 6 + m * 3 + p) * 64];
    }
    if (LN == 1) {          // the next layer's k-step 0 (second-to-last k-step) and k-step 1 (last k-step): no request of this layer left
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < NPC; ++p) nxt.w[m][p] = next_wp[(m * 3 + p) * 64];
    }
    if (LN == 2) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < NPC; ++p) nxt.w1[m][p] = next_wp[(6 + m * 3 + p) * 64];
    }
    if (LX) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int p = 0; p < NPC; ++p) r.x[R1][n][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP) + 32 * n * LDP + 16 * (t + 1));
    }
    // smallest terms first; four accumulators alternate
    // (the first product of a GEMM takes C = 0 as an inline constant: no accumulator zeroing)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 ci0 = zero16, ci1 = zero16;
    if constexpr (NT == 2) {
#define SPF_X3(PW, PX, Z)                                                                                              \
    {                                                                                                                  \
        const f32x16 c00 = (Z) ? ci0 : acc[0][0], c01 = (Z) ? ci0 : acc[0][1], c10 = (Z) ? ci1 : acc[1][0],                \
                     c11 = (Z) ? ci1 : acc[1][1];                                                                      \
        if (!SWAP) {                                                                                                   \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0][PW], r.x[R][0][PX], c00, 0, 0, 0);             \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0][PW], r.x[R][1][PX], c01, 0, 0, 0);             \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][1][PW], r.x[R][0][PX], c10, 0, 0, 0);             \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][1][PW], r.x[R][1][PX], c11, 0, 0, 0);             \
        } else {                                                                                                       \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.x[R][0][PX], r.w[R][0][PW], c00, 0, 0, 0);             \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.x[R][0][PX], r.w[R][1][PW], c01, 0, 0, 0);             \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.x[R][1][PX], r.w[R][0][PW], c10, 0, 0, 0);             \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.x[R][1][PX], r.w[R][1][PW], c11, 0, 0, 0);             \
        }                                                                                                              \
    }
#define SPF_H2(PW, PX, Z, A)                                                                                           \
    {                                                                                                                  \
        const f32x16 c00 = (Z) ? zero16 : A[0][0], c01 = (Z) ? zero16 : A[0][1], c10 = (Z) ? zero16 : A[1][0],             \
                     c11 = (Z) ? zero16 : A[1][1];                                                                     \
        const f16x8 w0_ = __builtin_bit_cast(f16x8, r.w[R][0][PW]), w1_ = __builtin_bit_cast(f16x8, r.w[R][1][PW]);      \
        const f16x8 x0_ = __builtin_bit_cast(f16x8, r.x[R][0][PX]), x1_ = __builtin_bit_cast(f16x8, r.x[R][1][PX]);      \
        if (!SWAP) {                                                                                                   \
            A[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0_, x0_, c00, 0, 0, 0);                                    \
            A[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0_, x1_, c01, 0, 0, 0);                                    \
            A[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1_, x0_, c10, 0, 0, 0);                                    \
            A[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1_, x1_, c11, 0, 0, 0);                                    \
        } else {                                                                                                       \
            A[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0_, w0_, c00, 0, 0, 0);                                    \
            A[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0_, w1_, c01, 0, 0, 0);                                    \
            A[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1_, w0_, c10, 0, 0, 0);                                    \
            A[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1_, w1_, c11, 0, 0, 0);                                    \
        }                                                                                                              \
    }
        if constexpr (H2) {         // cross terms into their own accumulator (scaled by 2^-11 once per layer), the main term into `acc`
            SPF_H2(1, 0, FIRST, accc) SPF_H2(0, 1, false, accc) SPF_H2(0, 0, FIRST, acc)
        } else if constexpr (NPC == 3) {
            SPF_X3(2, 0, FIRST) SPF_X3(0, 2, false) SPF_X3(1, 1, false) SPF_X3(1, 0, false) SPF_X3(0, 1, false) SPF_X3(0, 0, false)
        } else {
            SPF_X3(1, 0, FIRST) SPF_X3(0, 1, false) SPF_X3(0, 0, false)
        }
#undef SPF_H2
#undef SPF_X3
        // the requests ride between the MFMAs (issued in one block in front of them, their ~25 issue slots leave the matrix pipe idle
        // once per k-step): LDS reads first (needed at the start of the next k-step), then the L2 requests (needed one k-step later)
        if (LX) {
#pragma unroll
            for (int i = 0; i < 2 * NPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        if (LW || LN != 0) {
#pragma unroll
            for (int i = 0; i < 2 * NPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, NPC == 3 ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    } else {
        static_assert(NT == 1 && !SWAP, "half-height tiles: the transposed product only");
#define SPF_X3H(PW, PX, Z)                                                                                             \
    {                                                                                                                  \
        const f32x16 c0 = (Z) ? ci0 : acc[0][0], c1 = (Z) ? ci1 : acc[1][0];                                             \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0][PW], r.x[R][0][PX], c0, 0, 0, 0);                  \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][1][PW], r.x[R][0][PX], c1, 0, 0, 0);                  \
    }
#define SPF_H2H(PW, PX, Z, A)                                                                                          \
    {                                                                                                                  \
        const f32x16 c0 = (Z) ? zero16 : A[0][0], c1 = (Z) ? zero16 : A[1][0];                                           \
        const f16x8 x0_ = __builtin_bit_cast(f16x8, r.x[R][0][PX]);                                                     \
        A[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r.w[R][0][PW]), x0_, c0, 0, 0, 0);   \
        A[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r.w[R][1][PW]), x0_, c1, 0, 0, 0);   \
    }
        if constexpr (H2) {
            SPF_H2H(1, 0, FIRST, accc) SPF_H2H(0, 1, false, accc) SPF_H2H(0, 0, FIRST, acc)
        } else if constexpr (NPC == 3) {
            SPF_X3H(2, 0, FIRST) SPF_X3H(0, 2, false) SPF_X3H(1, 1, false) SPF_X3H(1, 0, false) SPF_X3H(0, 1, false) SPF_X3H(0, 0, false)
        } else {
            SPF_X3H(1, 0, FIRST) SPF_X3H(0, 1, false) SPF_X3H(0, 0, false)
        }
#undef SPF_H2H
#undef SPF_X3H
        // 12 MFMAs carry 3 LDS reads and 6 L2 requests
        if (LX) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        if (LW || LN != 0) {
#pragma unroll
            for (int i = 0; i < 2 * NPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int T, bool SWAP = false, int LDP = X3_LDP, int NT = 2, int NPC = 3, bool H2 = false>
__device__ __forceinline__ WFrag3 gemm_x3(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][NT], const WFrag3& first, gx3 next_wp,
                                          f32x16 (&accc)[2][NT]) {
    static_assert(T >= 3, "gemm_x3: at least three k-steps");
    const int j = lane & 31, kg = lane >> 5;
    const __bf16* xp = X + j * LDP + 8 * kg;
    X3Regs<NT> r;
    WFrag3 nxt = first;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
            r.w[0][m][p] = first.w[m][p];
            r.w[1][m][p] = first.w1[m][p];
        }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int p = 0; p < NPC; ++p) r.x[0][n][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP) + 32 * n * LDP);
    constexpr int MAIN = T - 2, REM = MAIN % 3;        // k-steps that request weights; the last two only consume
    constexpr bool ZC = true;                          // C = 0 on the first product of the GEMM
    if (MAIN >= 3) {
        x3_step<0, true, true, 0, SWAP, LDP, ZC, NT, NPC, H2>(xp, wp, 0, acc, r, nxt, next_wp, accc);
        x3_step<1, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, 1, acc, r, nxt, next_wp, accc);
        x3_step<2, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, 2, acc, r, nxt, next_wp, accc);
#pragma unroll 1
        for (int t = 3; t + 3 <= MAIN; t += 3) {
            x3_step<0, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, t, acc, r, nxt, next_wp, accc);
            x3_step<1, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, t + 1, acc, r, nxt, next_wp, accc);
            x3_step<2, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, t + 2, acc, r, nxt, next_wp, accc);
        }
    }
    if (REM >= 1) x3_step<0, true, true, 0, SWAP, LDP, (ZC && MAIN < 3), NT, NPC, H2>(xp, wp, MAIN - REM, acc, r, nxt, next_wp, accc);
    if (REM == 2) x3_step<1, true, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, MAIN - 1, acc, r, nxt, next_wp, accc);
    if (next_wp) {
        x3_step<REM, false, true, 1, SWAP, LDP, false, NT, NPC, H2>(xp, wp, T - 2, acc, r, nxt, next_wp, accc);
        x3_step<(REM + 1) % 3, false, false, 2, SWAP, LDP, false, NT, NPC, H2>(xp, wp, T - 1, acc, r, nxt, next_wp, accc);
    } else {
        x3_step<REM, false, true, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, T - 2, acc, r, nxt, next_wp, accc);
        x3_step<(REM + 1) % 3, false, false, 0, SWAP, LDP, false, NT, NPC, H2>(xp, wp, T - 1, acc, r, nxt, next_wp, accc);
    }
    return nxt;
}
// (the six- / three-product bf16 forms: one accumulator)
template <int T, bool SWAP = false, int LDP = X3_LDP, int NT = 2, int NPC = 3>
__device__ __forceinline__ WFrag3 gemm_x3(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][NT], const WFrag3& first, gx3 next_wp) {
    return gemm_x3<T, SWAP, LDP, NT, NPC, false>(X, wp, lane, acc, first, next_wp, acc);
}
// H2: acc = main + 2^-11 cross, the layer's result in the layout every epilogue expects
template <int NT>
__device__ __forceinline__ void h2_combine(f32x16 (&acc)[2][NT], const f32x16 (&accc)[2][NT]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][n][q] = __builtin_fmaf(accc[m][n][q], H2_EPS, acc[m][n][q]);
}

// One 32x32 output tile per wave, D[m-th 32 weight rows][n-th 32 rows of X] over T k16-steps (the narrow last products: 256 ->
// latent / input width).  wp: [T][3][64] fragments of the wave's weight rows, + lane.  `pre`: k-steps 0 and 1, requested by the
// caller ahead of the preceding epilogue.  Small terms and large terms go to separate accumulators (two independent MFMA chains).
struct WFrag1 {
    bf16x8 w[2][3];
};
__device__ __forceinline__ WFrag1 load_wfrag1(gx3 wp) {
    WFrag1 f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[s][p] = wp[(s * 3 + p) * 64];
    return f;
}
struct X1Regs {
    bf16x8 w[3][3], x[3][3];
};
template <int R, bool LW, bool LX, int LDP = X3_LDP, bool H2 = false>
__device__ __forceinline__ void x1_step(const __bf16* xp, gx3 wp, int t, f32x16& lo, f32x16& hi, X1Regs& r) {
    constexpr int NPC1 = H2 ? 2 : 3;
    constexpr int R1 = (R + 1) % 3, R2 = (R + 2) % 3;
    if (LW) {
#pragma unroll
        for (int p = 0; p < NPC1; ++p) r.w[R2][p] = wp[((t + 2) * 3 + p) * 64];
    }
    if (LX) {
#pragma unroll
        for (int p = 0; p < NPC1; ++p) r.x[R1][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP) + 16 * (t + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H2) {          // lo = cross terms (scaled by the caller), hi = main term
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r.w[R][1]), __builtin_bit_cast(f16x8, r.x[R][0]), lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r.w[R][0]), __builtin_bit_cast(f16x8, r.x[R][0]), hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r.w[R][0]), __builtin_bit_cast(f16x8, r.x[R][1]), lo, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][2], r.x[R][0], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][1], r.x[R][0], hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0], r.x[R][2], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0], r.x[R][1], hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][1], r.x[R][1], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.w[R][0], r.x[R][0], hi, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// returns the tile in the accumulator layout (row_of / column = lane & 31)
template <int T, int LDP = X3_LDP, bool H2 = false>
__device__ __forceinline__ f32x16 gemm_x3_tile(const __bf16* X, int n, gx3 wp, int lane, const WFrag1& pre) {
    static_assert(T >= 3 && (T - 2) % 3 != 0, "gemm_x3_tile: tail phases are written for (T - 2) mod 3 in {1, 2}");
    const int j = lane & 31, kg = lane >> 5;
    const __bf16* xp = X + (32 * n + j) * LDP + 8 * kg;
    X1Regs r;
    f32x16 lo, hi;
#pragma unroll
    for (int q = 0; q < 16; ++q) lo[q] = hi[q] = 0.f;
#pragma unroll
    for (int p = 0; p < (H2 ? 2 : 3); ++p) {
        r.w[0][p] = pre.w[0][p];
        r.w[1][p] = pre.w[1][p];
        r.x[0][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP));
    }
    constexpr int MAIN = T - 2, REM = MAIN % 3;
    int t = 0;
#pragma unroll 1
    for (; t + 3 <= MAIN; t += 3) {
        x1_step<0, true, true, LDP, H2>(xp, wp, t, lo, hi, r);
        x1_step<1, true, true, LDP, H2>(xp, wp, t + 1, lo, hi, r);
        x1_step<2, true, true, LDP, H2>(xp, wp, t + 2, lo, hi, r);
    }
    if (REM >= 1) x1_step<0, true, true, LDP, H2>(xp, wp, MAIN - REM, lo, hi, r);
    if (REM == 2) x1_step<1, true, true, LDP, H2>(xp, wp, MAIN - 1, lo, hi, r);
    x1_step<REM, false, true, LDP, H2>(xp, wp, T - 2, lo, hi, r);
    x1_step<(REM + 1) % 3, false, false, LDP, H2>(xp, wp, T - 1, lo, hi, r);
#pragma unroll
    for (int q = 0; q < 16; ++q) hi[q] = H2 ? __builtin_fmaf(lo[q], H2_EPS, hi[q]) : hi[q] + lo[q];
    return hi;
}

// ---- epilogue arithmetic, two elements per instruction where the ISA allows (v_cvt_pk_bf16_f32, v_pk_add_f32, v_pk_mul_f32) ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// (the compiler pairs the subtractions of the split on its own, but scalarises sums / products whose results feed inline asm)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// h = a + b and 0.01 h for four accumulator values
__device__ __forceinline__ void bias_scale4(const f32x16& acc, int g, f32x4 b, f32x4& h, f32x4& hs) {
    const f32x2 c = f32x2{0.01f, 0.01f};
    const f32x2 h0 = pk_add(f32x2{acc[4 * g], acc[4 * g + 1]}, f32x2{b[0], b[1]}), h1 = pk_add(f32x2{acc[4 * g + 2], acc[4 * g + 3]}, f32x2{b[2], b[3]});
    const f32x2 s0 = pk_mul(h0, c), s1 = pk_mul(h1, c);
    h = f32x4{h0[0], h0[1], h1[0], h1[1]};
    hs = f32x4{s0[0], s0[1], s1[0], s1[1]};
}
__device__ __forceinline__ void scale4(const f32x16& acc, int g, f32x4& v, f32x4& vs) {
    const f32x2 c = f32x2{0.01f, 0.01f};
    const f32x2 v0 = f32x2{acc[4 * g], acc[4 * g + 1]}, v1 = f32x2{acc[4 * g + 2], acc[4 * g + 3]};
    const f32x2 s0 = pk_mul(v0, c), s1 = pk_mul(v1, c);
    v = f32x4{v0[0], v0[1], v1[0], v1[1]};
    vs = f32x4{s0[0], s0[1], s1[0], s1[1]};
}

__device__ __forceinline__ uint32_t cvt_pk_bf16(f32x2 a) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)); }
__device__ __forceinline__ f32x2 pk_bf16_to_f32(uint32_t c) { return f32x2{__uint_as_float(c << 16), __uint_as_float(c & 0xffff0000u)}; }
// two floats -> three packed bf16 pairs, a = p1 + p2 + p3 exactly (same pieces as split3)
__device__ __forceinline__ void split3_pair(f32x2 a, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
    c1 = cvt_pk_bf16(a);
    const f32x2 r1 = a - pk_bf16_to_f32(c1);
    c2 = cvt_pk_bf16(r1);
    const f32x2 r2 = r1 - pk_bf16_to_f32(c2);
    c3 = cvt_pk_bf16(r2);
}

// H2: two floats -> (h1, h1') and (h2, h2') as packed fp16 pairs (round to nearest)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pair(f32x2 a, uint32_t& c1, uint32_t& c2) {
    const f16x2 h = __builtin_convertvector(a, f16x2);
    const f32x2 r = (a - __builtin_convertvector(h, f32x2)) * f32x2{2048.0f, 2048.0f};
    c1 = __builtin_bit_cast(uint32_t, h);
    c2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
// 4 consecutive features of one row: the two fp16 planes (the slots of bf16 pieces 0 and 1)
template <int LDP = X3_LDP>
__device__ __forceinline__ void store_quad_h2(__bf16* X, int row, int f0, f32x4 v) {
    uint32_t a1, a2, b1, b2;
    split2h_pair(f32x2{v[0], v[1]}, a1, a2);
    split2h_pair(f32x2{v[2], v[3]}, b1, b2);
    __bf16* dst = X + row * LDP + f0;
    *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(dst + (64 * LDP)) = u32x2{a2, b2};
}
template <bool H2, int LDP = X3_LDP>
__device__ __forceinline__ void store_quad_xh(__bf16* X, int row, int f0, f32x4 v);

// write 4 consecutive features of one row as three bf16 quads
template <int LDP = X3_LDP>
__device__ __forceinline__ void store_quad_x3(__bf16* X, int row, int f0, f32x4 v) {
    uint32_t a1, a2, a3, b1, b2, b3;
    split3_pair(f32x2{v[0], v[1]}, a1, a2, a3);
    split3_pair(f32x2{v[2], v[3]}, b1, b2, b3);
    __bf16* dst = X + row * LDP + f0;
    *reinterpret_cast<u32x2*>(dst) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(dst + (64 * LDP)) = u32x2{a2, b2};
    *reinterpret_cast<u32x2*>(dst + 2 * (64 * LDP)) = u32x2{a3, b3};
}
template <int LDP = X3_LDP>
__device__ __forceinline__ void store_quad_x3(__bf16* X, int row, int f0, const float (&v)[4]) {
    store_quad_x3<LDP>(X, row, f0, f32x4{v[0], v[1], v[2], v[3]});
}
template <bool H2, int LDP>
__device__ __forceinline__ void store_quad_xh(__bf16* X, int row, int f0, f32x4 v) {
    if constexpr (H2) store_quad_h2<LDP>(X, row, f0, v);
    else store_quad_x3<LDP>(X, row, f0, v);
}
template <bool H2, int LDP = X3_LDP>
__device__ __forceinline__ void store_quad_xh(__bf16* X, int row, int f0, const float (&v)[4]) {
    store_quad_xh<H2, LDP>(X, row, f0, f32x4{v[0], v[1], v[2], v[3]});
}

// LeakyReLU sign bits travel in 32-bit words filled from the top: push appends (h > 0) below the bits already there
// (bits = 2 bits + (h > 0), the carry-in form of the add), pop takes the top bit (bits = 2 bits, carry-out).  A word that received 32
// pushes pops them in the same order.  h001 = 0.01 h is formed outside (two per instruction).
__device__ __forceinline__ float lrelu_push(float h, float h001, uint32_t& bits) {
    float out;
    asm("v_cmp_lt_f32_e32 vcc, 0, %2\n\tv_cndmask_b32_e32 %0, %3, %2, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "=&v"(out), "+v"(bits)
        : "v"(h), "v"(h001)
        : "vcc");
    return out;
}
// same, and also selects `a` (h > 0) or `a001` into `sel` (the Jacobian seed of the last forward layer)
__device__ __forceinline__ float lrelu_push_sel(float h, float h001, float a, float a001, float& sel, uint32_t& bits) {
    float out;
    asm("v_cmp_lt_f32_e32 vcc, 0, %3\n\tv_cndmask_b32_e32 %0, %4, %3, vcc\n\tv_cndmask_b32_e32 %1, %6, %5, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "=&v"(out), "=&v"(sel), "+v"(bits)
        : "v"(h), "v"(h001), "v"(a), "v"(a001)
        : "vcc");
    return out;
}
__device__ __forceinline__ float lrelu_pop(float g, float g001, uint32_t& bits) {
    float out;
    asm("v_add_co_u32_e32 %1, vcc, %1, %1\n\tv_cndmask_b32_e32 %0, %3, %2, vcc" : "=&v"(out), "+v"(bits) : "v"(g), "v"(g001) : "vcc");
    return out;
}

// [ROWS][8 * NG] tile of the planes (exactly p1 + p2 + p3 per element) -> fp32 rows in HBM, coalesced (32 B per thread, a row's
// threads are consecutive): what the weight-gradient GEMM reads.
template <int NG, int LDP = X3_LDP, int ROWS = 64>
__device__ __forceinline__ void store_tile_from_planes(const __bf16* X, float* __restrict__ dst, int ld_dst, int tid) {
    for (int idx = tid; idx < ROWS * NG; idx += 256) {
        const int row = idx / NG, gc = idx % NG;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(X + row * LDP + 8 * gc);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(X + (64 * LDP) + row * LDP + 8 * gc);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(X + 2 * (64 * LDP) + row * LDP + 8 * gc);
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

// the same for either piece form; H2: h1 + 2^-11 h2 per element (the 22-bit value the next GEMM consumed), optionally times a per-row factor
// (block-scaled gradient rows: rinv[row] undoes the row's power of two, exactly)
template <bool H2, int NG, int LDP = X3_LDP, int ROWS = 64>
__device__ __forceinline__ void store_tile_from_planes_xh(const __bf16* X, float* __restrict__ dst, int ld_dst, int tid, const float* rinv = nullptr) {
    if constexpr (!H2) {
        store_tile_from_planes<NG, LDP, ROWS>(X, dst, ld_dst, tid);
    } else {
        for (int idx = tid; idx < ROWS * NG; idx += 256) {
            const int row = idx / NG, gc = idx % NG;
            const f16x8 a = *reinterpret_cast<const f16x8*>(X + row * LDP + 8 * gc);
            const f16x8 b = *reinterpret_cast<const f16x8*>(X + (64 * LDP) + row * LDP + 8 * gc);
            const float r = rinv ? rinv[row] : 1.0f;
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = __builtin_fmaf((float)b[e], H2_EPS, (float)a[e]) * r;
                hi[e] = __builtin_fmaf((float)b[e + 4], H2_EPS, (float)a[e + 4]) * r;
            }
            float* d = dst + (size_t)row * ld_dst + 8 * gc;
            *reinterpret_cast<f32x4*>(d) = lo;
            *reinterpret_cast<f32x4*>(d + 4) = hi;
        }
    }
}

// one element (row, col) of the planes
template <bool H2, int LDP = X3_LDP>
__device__ __forceinline__ void store_one_xh(__bf16* X, int row, int col, float v);
template <int LDP = X3_LDP>
__device__ __forceinline__ void store_one_x3(__bf16* X, int row, int col, float v) {
    __bf16 a, b, c;
    split3(v, a, b, c);
    X[row * LDP + col] = a;
    X[(64 * LDP) + row * LDP + col] = b;
    X[2 * (64 * LDP) + row * LDP + col] = c;
}
template <bool H2, int LDP>
__device__ __forceinline__ void store_one_xh(__bf16* X, int row, int col, float v) {
    if constexpr (H2) {
        const _Float16 h = (_Float16)v, g = (_Float16)((v - (float)h) * 2048.0f);
        X[row * LDP + col] = __builtin_bit_cast(__bf16, h);
        X[(64 * LDP) + row * LDP + col] = __builtin_bit_cast(__bf16, g);
    } else {
        store_one_x3<LDP>(X, row, col, v);
    }
}

}  // namespace spf
