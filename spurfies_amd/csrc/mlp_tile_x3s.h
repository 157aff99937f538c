// The bf16-piece ("x3") tile engine on v_mfma_f32_16x16x32_bf16 — same arithmetic as mlp_tile_x3.h (three bf16 pieces per operand, the
// six piece products with i + j <= 4, fp32 accumulate: fp32-class products (<= 2 ulp per product)), same operand traffic, same 64 features x 64 rows per
// wave, but 4 x 4 tiles of 16 x 16 with K = 32 per instruction instead of 2 x 2 tiles of 32 x 32 with K = 16.
// Why: under a dense matrix load the chip lowers its clock, and the clock it holds depends on the MFMA shape.  Measured with this
// loop's operand traffic on random data (tools/micro/mfma_shape_rate.hip, one 4-wave workgroup per CU, layers of 256 -> 256):
// 32x32x16 264.9 fp32-equivalent TFLOP/s at 1.79 GHz, 16x16x32 293.5 at 2.02 GHz (+10.8 % wall at +1.7 % cycles); on all-zero data both
// run at 2.37 GHz and the cycle ratio decides.
// Layout: transposed product D[feature][row] += W[feature][k] X[k][row].  A operand = weights, lane (i = lane & 15, g = lane >> 4):
// feature 16 a + i, k = 32 t + 8 g .. + 7, streamed from L2 as fragments [wave][k32][a][piece][lane] x 8 bf16.  B operand =
// activations from the LDS planes [piece][row][k]: row 16 b + i, same k.  Accumulator acc[a][b] (4 floats): features
// 64 wave + 16 a + 4 g + r, row 16 b + i — a lane owns 4 CONSECUTIVE features of 4 rows per feature tile, so the epilogues keep their
// 8-byte plane stores (store_quad_x3).
#pragma once
#include "mlp_tile_x3.h"

namespace spf {

struct WFragS {
    bf16x8 w[4][3];       // k32-step 0 of a layer
};
// this lane's bias values of a layer (features 64 wave + 16 a + 4 g .. + 3)
struct BiasS {
    f32x4 b[4];
};
__device__ __forceinline__ BiasS load_biass(gfp bias, int wave, int lane) {
    BiasS r;
#pragma unroll
    for (int a = 0; a < 4; ++a)
        r.b[a] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(bias + 64 * wave + 16 * a + 4 * (lane >> 4));
    return r;
}
__device__ __forceinline__ WFragS load_wfrags(gx3 wp) {
    WFragS f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[a][p] = wp[(a * 3 + p) * 64];
    return f;
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Operand registers of the k-loop.  Pieces 1 and 2 of both operands are SINGLE buffered: the six products of a k-step run in the order
// (w2 x0) (w0 x2) (w1 x1) (w1 x0) (w0 x1) (w0 x0), so w2 is free after the first sixteen MFMAs, x2 after the second sixteen, w1 after
// the fourth, x1 after the fifth — the next k-step's fragments are requested into the same registers right there, 48 - 80 MFMAs
// (768 - 1280 cycles) before their first use.  Piece 0 is used by the LAST product of a k-step and the first two of the next one, so it
// alone is double buffered.  32 fragments = 128 registers (a fully double-buffered set is 192, and with the 64 accumulators the
// allocator then shuffles fragments through the AGPRs every iteration: ~115 copies per two k-steps, measured as 13 % of the loop).
struct XSRegs {
    bf16x8 w0[2][4], w1[4], w2[4], x0[2][4], x1[4], x2[4];
};

#define SPF_XS_GROUP(WV, XV)                                                                                                 \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b)                            \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WV[a], XV[b], acc[a][b], 0, 0, 0);
// NV x (NM MFMAs, one request of class MASK): 0x020 = vector-memory read, 0x100 = LDS read
#define SPF_XS_PIN(NV, NM, MASK)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {                                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);                                                                    \
    }

// One k32-step: this k-step's 96 MFMAs (smallest terms first; sixteen accumulators alternate) with the requests of k-step t + 1 riding
// between them (LD: there is a k-step t + 1; LN: last k-step, request the NEXT layer's k-step 0; NB: last k-step, request NB quads of
// epilogue constants per feature tile from `cptr`).  C = which piece-0 buffer holds k-step t.
template <int C, bool LD, bool LN, int NB, int LDP = X3_LDP>
__device__ __forceinline__ void xs_step(const __bf16* xp, gx3 wp, int t, f32x4 (&acc)[4][4], XSRegs& r, WFragS& nxt, gx3 next_wp,
                                        gfp cptr0, gfp cptr1, BiasS& c0, BiasS& c1) {
    constexpr int N = C ^ 1;
    const __bf16* xn = xp + 32 * (t + 1);
    gx3 wn = wp + (size_t)(t + 1) * 12 * 64;
    if (LD) {
#pragma unroll
        for (int a = 0; a < 4; ++a) r.w0[N][a] = wn[(a * 3 + 0) * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b) r.x0[N][b] = *reinterpret_cast<const bf16x8*>(xn + 16 * b * LDP);
    }
    if (LN) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) nxt.w[a][p] = next_wp[(a * 3 + p) * 64];
    }
    if (NB >= 1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) c0.b[a] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(cptr0 + 16 * a);
    }
    if (NB >= 2) {
#pragma unroll
        for (int a = 0; a < 4; ++a) c1.b[a] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(cptr1 + 16 * a);
    }
    SPF_XS_GROUP(r.w2, r.x0[C])
    if (LD) {
#pragma unroll
        for (int a = 0; a < 4; ++a) r.w2[a] = wn[(a * 3 + 2) * 64];
    }
    SPF_XS_GROUP(r.w0[C], r.x2)
    if (LD) {
#pragma unroll
        for (int b = 0; b < 4; ++b) r.x2[b] = *reinterpret_cast<const bf16x8*>(xn + 2 * (64 * LDP) + 16 * b * LDP);
    }
    SPF_XS_GROUP(r.w1, r.x1)
    SPF_XS_GROUP(r.w1, r.x0[C])
    if (LD) {
#pragma unroll
        for (int a = 0; a < 4; ++a) r.w1[a] = wn[(a * 3 + 1) * 64];
    }
    SPF_XS_GROUP(r.w0[C], r.x1)
    if (LD) {
#pragma unroll
        for (int b = 0; b < 4; ++b) r.x1[b] = *reinterpret_cast<const bf16x8*>(xn + (64 * LDP) + 16 * b * LDP);
    }
    SPF_XS_GROUP(r.w0[C], r.x0[C])
    if (LD) {
        SPF_XS_PIN(4, 2, 0x020) SPF_XS_PIN(4, 2, 0x100)       // products 1: next w0, next x0
        SPF_XS_PIN(4, 4, 0x020)                               // product 2: next w2
        SPF_XS_PIN(4, 4, 0x100)                               // product 3: next x2
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // product 4
        SPF_XS_PIN(4, 4, 0x020)                               // product 5: next w1
        SPF_XS_PIN(4, 4, 0x100)                               // product 6: next x1
    }
    if (LN || NB > 0) { SPF_XS_PIN((LN ? 12 : 0) + 4 * NB, 4, 0x020) }
    __builtin_amdgcn_sched_barrier(0);
}

// acc += W X over T k32-steps (T even).  wp: this wave's fragments of the layer, + lane.  `first`: the layer's k-step-0 fragments,
// requested by the previous layer's GEMM (its last k-step) so that their L2 round trip is not exposed behind the barriers; `nxt`
// receives the k-step-0 fragments of `next_wp` (HAS_NEXT), c0 / c1 the epilogue's per-feature constants (bias, folded last layer),
// requested in the last k-step as well: nothing that only the epilogue needs occupies registers during the k-loop.
template <int T, bool HAS_NEXT, int NB, int LDP = X3_LDP>
__device__ __forceinline__ void gemm_xs(const __bf16* X, gx3 wp, int lane, f32x4 (&acc)[4][4], const WFragS& first, WFragS& nxt, gx3 next_wp,
                                        gfp cptr0, gfp cptr1, BiasS& c0, BiasS& c1) {
    static_assert(T >= 2 && T % 2 == 0, "gemm_xs: an even number of k32-steps");
    const int i = lane & 15, g = lane >> 4;
    const __bf16* xp = X + i * LDP + 8 * g;
    XSRegs r;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        r.w0[0][a] = first.w[a][0];
        r.w1[a] = first.w[a][1];
        r.w2[a] = first.w[a][2];
        r.x0[0][a] = *reinterpret_cast<const bf16x8*>(xp + 16 * a * LDP);
        r.x2[a] = *reinterpret_cast<const bf16x8*>(xp + 2 * (64 * LDP) + 16 * a * LDP);
        r.x1[a] = *reinterpret_cast<const bf16x8*>(xp + (64 * LDP) + 16 * a * LDP);
    }
#pragma unroll       // straight-line: as a loop, the back edge permutes the sixteen accumulators through ~110 register copies per iteration
    for (int t = 0; t + 2 < T; t += 2) {
        xs_step<0, true, false, 0, LDP>(xp, wp, t, acc, r, nxt, next_wp, cptr0, cptr1, c0, c1);
        xs_step<1, true, false, 0, LDP>(xp, wp, t + 1, acc, r, nxt, next_wp, cptr0, cptr1, c0, c1);
    }
    xs_step<0, true, false, 0, LDP>(xp, wp, T - 2, acc, r, nxt, next_wp, cptr0, cptr1, c0, c1);
    xs_step<1, false, HAS_NEXT, NB, LDP>(xp, wp, T - 1, acc, r, nxt, next_wp, cptr0, cptr1, c0, c1);
}

#undef SPF_XS_GROUP
#undef SPF_XS_PIN

// 32 weight rows x 32 rows of X per wave (2 x 2 tiles) over T k32-steps: the narrow last products (256 -> latent / input width).
// wp: [T][2][3][64] fragments of the wave's weight rows, + lane; rows 32 n .. 32 n + 31 of X.
struct WFragS2 {
    bf16x8 w[2][3];
};
__device__ __forceinline__ WFragS2 load_wfrags2(gx3 wp) {
    WFragS2 f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) f.w[a][p] = wp[(a * 3 + p) * 64];
    return f;
}
template <int T, int LDP = X3_LDP>
__device__ __forceinline__ void gemm_xs_tile(const __bf16* X, int n, gx3 wp, int lane, const WFragS2& pre, f32x4 (&acc)[2][2]) {
    const int i = lane & 15, g = lane >> 4;
    const __bf16* xp = X + (32 * n + i) * LDP + 8 * g;
    bf16x8 w[2][2][3], x[2][2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            w[0][a][p] = pre.w[a][p];
            x[0][a][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP) + 16 * a * LDP);
        }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int c = t & 1, nx = c ^ 1;
        if (t + 1 < T) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    w[nx][a][p] = wp[((t + 1) * 6 + a * 3 + p) * 64];
                    x[nx][a][p] = *reinterpret_cast<const bf16x8*>(xp + p * (64 * LDP) + 16 * a * LDP + 32 * (t + 1));
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#define SPF_XS2(PW, PX)                                                                                                     \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)                            \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][a][PW], x[c][b][PX], acc[a][b], 0, 0, 0);
        SPF_XS2(2, 0) SPF_XS2(0, 2) SPF_XS2(1, 1) SPF_XS2(1, 0) SPF_XS2(0, 1) SPF_XS2(0, 0)
#undef SPF_XS2
        __builtin_amdgcn_sched_barrier(0);
    }
}

// h = a + b and 0.01 h for the four values of one accumulator
__device__ __forceinline__ void bias_scale4s(const f32x4& acc, f32x4 b, f32x4& h, f32x4& hs) {
    const f32x2 c = f32x2{0.01f, 0.01f};
    const f32x2 h0 = pk_add(f32x2{acc[0], acc[1]}, f32x2{b[0], b[1]}), h1 = pk_add(f32x2{acc[2], acc[3]}, f32x2{b[2], b[3]});
    const f32x2 s0 = pk_mul(h0, c), s1 = pk_mul(h1, c);
    h = f32x4{h0[0], h0[1], h1[0], h1[1]};
    hs = f32x4{s0[0], s0[1], s1[0], s1[1]};
}
__device__ __forceinline__ void scale4s(const f32x4& acc, f32x4& vs) {
    const f32x2 c = f32x2{0.01f, 0.01f};
    const f32x2 s0 = pk_mul(f32x2{acc[0], acc[1]}, c), s1 = pk_mul(f32x2{acc[2], acc[3]}, c);
    vs = f32x4{s0[0], s0[1], s1[0], s1[1]};
}

}  // namespace spf
