// Head stage (K5b of DESIGN.md), per valid POINT: agg = F_color.6(agg3) — the linear last layer of F_color, applied after
// the RBF-weighted mean it commutes with (color_mlp.hip) — then the radiance head `R`:
// [dir-enc3(ray dir) | agg] (277) -> 256 -> 256 -> 3 -> sigmoid; forward and the data-gradient chain of the backward,
// on the fp32 matrix cores.
//
// Replaces F_color.6 and the second half of get_color, spurfies/model/pointneus_disent.py:333-346 (view encoding
// embedder.py:26-30 with multires 3, torch.concat, cuBLAS GEMMs, sigmoid) and autograd's backward.
//
// Tile = 64 consecutive valid points (rows).  Internal column order is [agg (256) | dir-enc (21) | pad] so that the
// agg rows load 16-B aligned; spf_rhead_pack folds the permutation into the packed first-layer weights.
// Weight gradients of the 256-wide layers are GEMMs over stored [P,256] buffers (G_l^T act_{l-1}: spf_wgrad); the
// 3 x 256 last layer's and all bias gradients are accumulated in-kernel.
#include "mlp_tile.h"
#include "mlp_tile_x3.h"
#include "rhead_pack.h"

namespace {
using namespace spf;

template <bool STORE>
__device__ __forceinline__ void r_fwd_epilogue(float* X, const f32x16 (&acc)[2][2], const float (&bv)[2], int wave, int lane,
                                               uint32_t* mask_g) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        uint32_t bits = 0u;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] + bv[n];
                const bool pos = v > 0.f;
                bits |= (pos ? 1u : 0u) << (n * 16 + r);
                v = pos ? v : v * 0.01f;
                const int row = m * 32 + row_of(r, h);
                X[row * LDR + c0 + 32 * n] = v;
            }
        if (STORE) mask_g[(wave * 2 + m) * 64 + lane] = bits;
    }
}

template <bool STORE>
__global__ void __launch_bounds__(256, 2)
rhead_forward_kernel(const float* __restrict__ agg3, const float* __restrict__ ray_dirs, const int32_t* __restrict__ point_slot,
                     const int32_t* __restrict__ n_points_dev, int max_points, int SR, const float* packed, float* __restrict__ colors,
                     float* __restrict__ agg, float* __restrict__ direnc, float* __restrict__ act1, float* __restrict__ act2,
                     uint32_t* __restrict__ masks) {
    __shared__ __attribute__((aligned(16))) float smem[RL_TOTAL];
    float* X = smem + RL_X;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = n_points_dev ? min(*n_points_dev, max_points) : max_points;
    const int ntiles = (P + 63) / 64;
    const float* packed0 = packed;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp packed = launder(packed0);
        gf4p pk4 = reinterpret_cast<gf4p>(packed);
        const BFrag fr6 = load_bfrag(pk4 + (RO_FW6 / 4) + wave * (T_HID * 128), lane);     // in flight during the gather
        {   // gather: thread = (row, quarter): 64 agg3 floats each; quarter 0 also encodes the view direction
            const int row = tid >> 2, q4 = tid & 3;
            const int p = tile * 64 + row;
            const bool ok = p < P;
            const f32x4* src = reinterpret_cast<const f32x4*>(agg3 + (size_t)(ok ? p : 0) * 256 + q4 * 64);
#pragma unroll 4
            for (int u = 0; u < 16; ++u) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) v = src[u];
                *reinterpret_cast<f32x4*>(X + row * LDR + q4 * 64 + 4 * u) = v;
            }
            if (q4 == 0) {
                float* e = X + row * LDR + 256;
                int srow = -1;
                if (ok) {
                    srow = point_slot ? point_slot[p] : p;
                    const float* dv = ray_dirs + (size_t)(srow / SR) * 3;
                    const float d[3] = {dv[0], dv[1], dv[2]};
                    e[0] = d[0]; e[1] = d[1]; e[2] = d[2];
                    float fr = 1.f;
#pragma unroll
                    for (int l = 0; l < DIR_FREQ; ++l) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float a = d[c] * fr;
                            e[3 + 6 * l + c] = sinf(a);
                            e[6 + 6 * l + c] = cosf(a);
                        }
                        fr *= 2.f;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 21; ++c) e[c] = 0.f;
                }
                e[21] = 0.f; e[22] = 0.f; e[23] = 0.f;
                if (STORE) {
#pragma unroll
                    for (int c4 = 0; c4 < 6; ++c4)
                        *reinterpret_cast<f32x4*>(direnc + (size_t)(tile * 64 + row) * 24 + 4 * c4) = *reinterpret_cast<const f32x4*>(e + 4 * c4);
                }
                reinterpret_cast<int*>(smem + RL_ROW)[row] = srow;
            }
        }
        __syncthreads();
        const size_t tb = (size_t)tile * 64 * 256;
        uint32_t* mk = STORE ? masks + (size_t)tile * 2 * 512 : nullptr;
        f32x16 acc[2][2];
        const int cb = wave * 64 + (lane & 31);
        gf4p wfw6 = pk4 + (RO_FW6 / 4) + wave * (T_HID * 128);
        gf4p wfw1 = pk4 + (RO_FW1 / 4) + wave * (T_RIN * 128);
        gf4p wfw2 = pk4 + (RO_FW2 / 4) + wave * (T_HID * 128);
        float bv[2] = {packed[RO_B6 + cb], packed[RO_B6 + cb + 32]};       // requested before the GEMM that needs them
        zero_acc(acc);
        BFrag nf = gemm_rows64<T_HID, LDR>(X, wfw6, lane, acc, fr6, wfw1);     // F_color.6 on the weighted mean
        __syncthreads();
        {   // agg = acc + b6 (linear) into columns 0..255 of X (the dir-enc columns stay); kept for R.0's weight gradient
            const int c0 = cb, h = lane >> 5;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m * 32 + row_of(r, h);
                        X[row * LDR + c0 + 32 * n] = acc[m][n][r] + bv[n];
                    }
        }
        __syncthreads();
        bv[0] = packed[RO_B1 + cb]; bv[1] = packed[RO_B1 + cb + 32];
        zero_acc(acc);
        nf = gemm_rows64<T_RIN, LDR>(X, wfw1, lane, acc, nf, wfw2, side_tile(STORE ? agg + tb : nullptr, tid));
        __syncthreads();
        r_fwd_epilogue<STORE>(X, acc, bv, wave, lane, mk);
        __syncthreads();
        bv[0] = packed[RO_B2 + cb]; bv[1] = packed[RO_B2 + cb + 32];
        zero_acc(acc);
        gemm_rows64<T_HID, LDR>(X, wfw2, lane, acc, nf, nullptr, side_tile(STORE ? act1 + tb : nullptr, tid));
        __syncthreads();
        r_fwd_epilogue<STORE>(X, acc, bv, wave, lane, STORE ? mk + 512 : nullptr);
        __syncthreads();
        if (STORE) store_tile_256<LDR>(X, act2 + tb, tid);
        {   // 256 -> 3 + sigmoid: 4 threads per row, interleaved float4 chunks
            const int row = tid >> 2, q4 = tid & 3;
            gf4p w3 = pk4 + RO_W3 / 4;
            float s[3] = {0.f, 0.f, 0.f};
#pragma unroll 2
            for (int mth = 0; mth < 16; ++mth) {
                const int c4 = q4 + 4 * mth;
                const f32x4 a = *reinterpret_cast<const f32x4*>(X + row * LDR + 4 * c4);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 v = w3[c * 64 + c4];
                    s[c] += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
                }
            }
            const int srow = reinterpret_cast<const int*>(smem + RL_ROW)[row];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s[c] += __shfl_xor(s[c], 1);
                s[c] += __shfl_xor(s[c], 2);
                if (q4 == 0 && srow >= 0) colors[(size_t)srow * 3 + c] = 1.f / (1.f + expf(-(s[c] + packed[RO_B3 + c])));
            }
        }
        __syncthreads();
    }
}

// backward epilogue: G_l = g_a * lrelu'(h_l); write X and G_l, add the column sums to the bias gradient
__device__ __forceinline__ void r_bwd_epilogue(float* X, const f32x16 (&acc)[2][2], int wave, int lane, const uint32_t (&mbits)[2],
                                               float* __restrict__ g_bias) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
    float cs[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const uint32_t bits = mbits[m];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m * 32 + row_of(r, h);
                float v = acc[m][n][r];
                v = ((bits >> (n * 16 + r)) & 1u) ? v : v * 0.01f;
                X[row * LDR + c0 + 32 * n] = v;
                cs[n] += v;
            }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float t = cs[n] + __shfl_xor(cs[n], 32);
        if (h == 0) atomicAdd(&g_bias[c0 + 32 * n], t);
    }
}

__global__ void __launch_bounds__(256, 2)
rhead_backward_kernel(const float* __restrict__ g_colors, const float* __restrict__ colors, const int32_t* __restrict__ point_slot,
                      const int32_t* __restrict__ n_points_dev, int max_points, const float* packed, const float* __restrict__ act2,
                      const uint32_t* __restrict__ masks, float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ g_agg,
                      float* __restrict__ g_agg3, float* __restrict__ g_b6, float* __restrict__ g_b0, float* __restrict__ g_b2, float* __restrict__ g_w4 /* [3,256] */, float* __restrict__ g_b4 /* [3] */) {
    __shared__ __attribute__((aligned(16))) float smem[RL_TOTAL];
    float* X = smem + RL_X;
    float* s_g3 = smem + RL_G3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = n_points_dev ? min(*n_points_dev, max_points) : max_points;
    const int ntiles = (P + 63) / 64;
    const float* packed0 = packed;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp packed = launder(packed0);
        gf4p pk4 = reinterpret_cast<gf4p>(packed);
        const size_t tb = (size_t)tile * 64 * 256;
        const BFrag fr2 = load_bfrag(pk4 + (RO_BW2 / 4) + wave * (T_HID * 128), lane);     // in flight during the small last-layer stage
        if (tid < 64) {   // dL/d(pre-sigmoid) = g_c * c (1 - c)
            const int p = tile * 64 + tid;
            float g[3] = {0.f, 0.f, 0.f};
            if (p < P) {
                const int srow = point_slot ? point_slot[p] : p;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float cc = colors[(size_t)srow * 3 + c];
                    g[c] = g_colors[(size_t)srow * 3 + c] * cc * (1.f - cc);
                }
            }
            s_g3[tid * 4] = g[0]; s_g3[tid * 4 + 1] = g[1]; s_g3[tid * 4 + 2] = g[2]; s_g3[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        {   // dW3[c][col] += sum_rows g3[row][c] a2[row][col]; db3[c] += sum_rows g3[row][c]   (thread = column)
            float a0 = 0.f, a1 = 0.f, a2v = 0.f;
            const int rows_here = min(64, P - tile * 64);
            for (int row = 0; row < rows_here; ++row) {
                const float a = act2[tb + row * 256 + tid];
                a0 += s_g3[row * 4] * a;
                a1 += s_g3[row * 4 + 1] * a;
                a2v += s_g3[row * 4 + 2] * a;
            }
            atomicAdd(&g_w4[tid], a0);
            atomicAdd(&g_w4[256 + tid], a1);
            atomicAdd(&g_w4[512 + tid], a2v);
            if (tid < 3) {
                float s = 0.f;
                for (int row = 0; row < 64; ++row) s += s_g3[row * 4 + tid];
                atomicAdd(&g_b4[tid], s);
            }
        }
        const uint32_t* mk = masks + (size_t)tile * 2 * 512;
        {   // G2 = (g3 W3) * lrelu'(h2), formed directly in accumulator layout
            const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
            float w3[2][3];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int c = 0; c < 3; ++c) w3[n][c] = packed[RO_W3 + c * 256 + c0 + 32 * n];
            float cs[2] = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint32_t bits = mk[512 + (wave * 2 + m) * 64 + lane];
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int row = m * 32 + row_of(r, h);
                    const f32x4 g3 = *reinterpret_cast<const f32x4*>(s_g3 + row * 4);       // one LDS read per row, both column halves
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        float v = g3[0] * w3[n][0] + g3[1] * w3[n][1] + g3[2] * w3[n][2];
                        v = ((bits >> (n * 16 + r)) & 1u) ? v : v * 0.01f;
                        X[row * LDR + c0 + 32 * n] = v;
                        cs[n] += v;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float t = cs[n] + __shfl_xor(cs[n], 32);
                if (h == 0) atomicAdd(&g_b2[c0 + 32 * n], t);
            }
        }
        __syncthreads();
        f32x16 acc[2][2];
        gf4p wbwa = pk4 + (RO_BWA / 4) + wave * (T_HID * 128);
        gf4p wbw6 = pk4 + (RO_BW6 / 4) + wave * (T_HID * 128);
        const uint32_t mb1[2] = {mk[(wave * 2) * 64 + lane], mk[(wave * 2 + 1) * 64 + lane]};
        zero_acc(acc);
        BFrag nf = gemm_rows64<T_HID, LDR>(X, pk4 + (RO_BW2 / 4) + wave * (T_HID * 128), lane, acc, fr2, wbwa, side_tile(G2 + tb, tid));
        __syncthreads();
        r_bwd_epilogue(X, acc, wave, lane, mb1, g_b0);
        __syncthreads();
        zero_acc(acc);
        nf = gemm_rows64<T_HID, LDR>(X, wbwa, lane, acc, nf, wbw6, side_tile(G1 + tb, tid));
        __syncthreads();
        {   // g_agg[p][col] -> HBM (operand of F_color.6's weight gradient; padded to whole tiles: no bounds test) and X;
            // its column sums are F_color.6's bias gradient
            const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
            float cs[2] = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m * 32 + row_of(r, h);
                        const float v = acc[m][n][r];
                        X[row * LDR + c0 + 32 * n] = v;
                        cs[n] += v;
                    }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float t = cs[n] + __shfl_xor(cs[n], 32);
                if (h == 0) atomicAdd(&g_b6[c0 + 32 * n], t);
            }
        }
        __syncthreads();
        zero_acc(acc);
        gemm_rows64<T_HID, LDR>(X, wbw6, lane, acc, nf, nullptr, side_tile(g_agg + tb, tid));     // g_agg3 = g_agg W6
        __syncthreads();
        {
            const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) X[(m * 32 + row_of(r, h)) * LDR + c0 + 32 * n] = acc[m][n][r];
        }
        __syncthreads();
        store_tile_256<LDR>(X, g_agg3 + tb, tid);
        __syncthreads();
    }
}


// ==============================================================================================================================
// The same two kernels on the bf16 matrix pipe with fp32-class products (<= 2 ulp per product) from three bf16 pieces per operand (arith = SPF_ARITH_SPLIT) or, since round 6
// and by default, three exact fp16 piece products from two fp16 pieces (arith = SPF_ARITH_H2: template flag H2); arith = SPF_ARITH_F32 selects the fp32-MFMA kernels above.  Engine: mlp_tile_x3.h; plane row stride 296 bf16 so that the first
// head layer's 288-wide input ([agg 256 | dir-enc 21 | 0]) is one GEMM.  One workgroup per CU (113.7 KB of planes).
// Bias gradients of the three 256-wide layers are column sums of g_agg / G1 / G2 and come from spf_wgrad (dbias) in this mode.
// ==============================================================================================================================
// (layout constants RX_* and the fragment packing: rhead_pack.h)


struct RxBias {
    f32x4 b[2][4];
};
__device__ __forceinline__ RxBias rx_load_bias(gfp bias, int wave, int lane) {
    RxBias r;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            r.b[m][g] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(bias + 64 * wave + 32 * m + 8 * g + 4 * (lane >> 5));
    return r;
}

// transposed epilogues (lane = row j of half n, 4 consecutive features per quad): linear (acc + b), LeakyReLU forward (sign words
// pushed in the order (m, g, e), stored lane-major) and LeakyReLU backward (popped in the same order)
// a layer's GEMM on the head kernels' engine: bf16 x 3 (one accumulator) or H2 (main + cross accumulators, combined here)
template <int T, int NT, bool H2, int NC>
__device__ __forceinline__ WFrag3 rx_gemm(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][NT], const WFrag3& first, gx3 next_wp, f32x16 (&accc)[2][NC]) {
    if constexpr (H2) {
        static_assert(NC == NT, "H2: one cross accumulator per main accumulator");
        const WFrag3 nf = gemm_x3<T, false, RX_LDP, NT, 2, true>(X, wp, lane, acc, first, next_wp, accc);
        h2_combine<NT>(acc, accc);
        return nf;
    } else {
        return gemm_x3<T, false, RX_LDP, NT>(X, wp, lane, acc, first, next_wp);
    }
}

template <int NT, bool H2 = false>
__device__ __forceinline__ void rx_linear_epilogue(__bf16* X, const f32x16 (&acc)[2][NT], const RxBias* bias, int wave, int lane) {
    const int j = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 h = f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                if (bias) h += bias->b[m][g];
                store_quad_xh<H2, RX_LDP>(X, 32 * n + j, f0, h);
            }
        }
}
template <bool STORE, int NT, bool H2 = false>
__device__ __forceinline__ void rx_fwd_epilogue(__bf16* X, const f32x16 (&acc)[2][NT], const RxBias& bias, int wave, int lane, uint32_t* masks_l) {
    const int j = lane & 31, kg = lane >> 5;
    uint32_t bits[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bits[n] = 0u;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 h, hs, out;
                bias_scale4(acc[m][n], g, bias.b[m][g], h, hs);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = lrelu_push(h[e], hs[e], bits[n]);
                store_quad_xh<H2, RX_LDP>(X, 32 * n + j, f0, out);
            }
        }
    if (STORE) {
#pragma unroll
        for (int n = 0; n < NT; ++n) masks_l[(NT * wave + n) * 64 + lane] = bits[n];
    }
}
template <int NT, bool H2 = false>
__device__ __forceinline__ void rx_bwd_epilogue(__bf16* X, const f32x16 (&acc)[2][NT], int wave, int lane, const uint32_t (&mw)[NT]) {
    const int j = lane & 31, kg = lane >> 5;
    uint32_t bits[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bits[n] = mw[n];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 v, vs, out;
                scale4(acc[m][n], g, v, vs);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = lrelu_pop(v[e], vs[e], bits[n]);
                store_quad_xh<H2, RX_LDP>(X, 32 * n + j, f0, out);
            }
        }
}

// Points covered by whole rounds of 64-point tiles on a grid of G workgroups (forward and backward split the list at the same place: the
// LeakyReLU sign words are stored per tile in the layout of the tile height that wrote them).
__device__ __forceinline__ int rx_full_rounds_points(int P, int G) { return (P / 64 / G) * G * 64; }

// One workgroup's tiles.  NT = 2: tiles of 64 points (rounds 1 - 4); NT = 1 (round 5): HALF-HEIGHT tiles of 32 points — twice the tiles at ~55 % of a
// tile's time each, taken when the launch has at most 32 points per workgroup (the 128-rays-per-GPU step: 6.5 k points were 102 tiles on 102 of 256
// CUs, one ~45 us tile pass each).  Forward and backward make the same choice (same point count, same grid): the sign words' layout depends on it.
template <bool STORE, int NT, bool H2 = false>
__device__ __forceinline__ void rhead_forward_x3_body(__bf16* X, float* red, int* s_row, const float* __restrict__ agg3, const float* __restrict__ ray_dirs,
                                                      const int32_t* __restrict__ point_slot, const int p0, const int P, int SR, const float* packed,
                                                      float* __restrict__ colors, float* __restrict__ agg, float* __restrict__ direnc,
                                                      float* __restrict__ act1, float* __restrict__ act2, uint32_t* __restrict__ masks) {
    // points [p0, P) of the list, p0 a multiple of 64: tile t covers points p0 + t ROWS ..
    constexpr int ROWS = 32 * NT, PARTS = 256 / ROWS, PW = 256 / PARTS, NV = PW / 4;      // gather: thread = (row, part): PW agg3 floats each
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (P - p0 + ROWS - 1) / ROWS;
    const float* packed0 = packed;
    T_DECL
    // this thread's gather operands (row = tid / PARTS, part = tid % PARTS; part 0 also the ray direction) are requested one
    // tile ahead: the slot lookup before the second GEMM, the rows right after the last GEMM's weight requests (vector-memory results
    // return in order: requested earlier, the GEMMs' weight fragments would wait behind these HBM rows; and a dependent lookup issued
    // after them would wait for all of them), so they land under the last epilogue and the stores
    f32x4 v[NV];
    float dnext[3] = {0.f, 0.f, 0.f};
    int srow_next = -1, slot_pf = -1;
    auto fetch_slot = [&](int t) {
        const int p = p0 + t * ROWS + tid / PARTS;
        slot_pf = ((tid % PARTS) == 0 && t < ntiles && p < P) ? (point_slot ? point_slot[p] : p) : -1;
    };
    auto fetch_rows = [&](int t) {
        const int row = tid / PARTS, q = tid % PARTS;
        const int p = p0 + t * ROWS + row;
        const bool ok = t < ntiles && p < P;
        srow_next = slot_pf;
        if (srow_next >= 0) {
            const float* dv = ray_dirs + (size_t)(srow_next / SR) * 3;
            dnext[0] = dv[0]; dnext[1] = dv[1]; dnext[2] = dv[2];
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(agg3 + (size_t)(ok ? p : 0) * 256 + q * PW);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u] = ok ? src[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    fetch_slot((int)blockIdx.x);
    fetch_rows((int)blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + (H2 ? RH_OFF : R_PACKED));
        gx3 w_fw6 = frag + RX_FW6 + wave * (RX_TH * 2 * 3 * 64) + lane;
        gx3 w_fw1 = frag + RX_FW1 + wave * (RX_T1 * 2 * 3 * 64) + lane;
        gx3 w_fw2 = frag + RX_FW2 + wave * (RX_TH * 2 * 3 * 64) + lane;
        const WFrag3 fr6 = load_wfrag3(w_fw6);               // in flight during the gather
        T_MARK(0)
        {   // gather: thread = (row, part): PW agg3 floats each; part 0 also encodes the view direction (columns 256..287)
            const int row = tid / PARTS, q = tid % PARTS;
#pragma unroll
            for (int u = 0; u < NV; ++u) store_quad_xh<H2, RX_LDP>(X, row, q * PW + 4 * u, v[u]);
            if (q == 0) {
                float e[32];
#pragma unroll
                for (int c = 0; c < 32; ++c) e[c] = 0.f;
                const int srow = srow_next;
                if (srow >= 0) {
                    const float d[3] = {dnext[0], dnext[1], dnext[2]};
                    e[0] = d[0]; e[1] = d[1]; e[2] = d[2];
                    float fr = 1.f;
#pragma unroll
                    for (int l = 0; l < DIR_FREQ; ++l) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float a = d[c] * fr;
                            e[3 + 6 * l + c] = sinf(a);
                            e[6 + 6 * l + c] = cosf(a);
                        }
                        fr *= 2.f;
                    }
                }
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) store_quad_xh<H2, RX_LDP>(X, row, 256 + 4 * c4, f32x4{e[4 * c4], e[4 * c4 + 1], e[4 * c4 + 2], e[4 * c4 + 3]});
                if (STORE) {
#pragma unroll
                    for (int c4 = 0; c4 < 6; ++c4)
                        *reinterpret_cast<f32x4*>(direnc + (size_t)(p0 + tile * ROWS + row) * 24 + 4 * c4) = f32x4{e[4 * c4], e[4 * c4 + 1], e[4 * c4 + 2], e[4 * c4 + 3]};
                }
                s_row[row] = srow;
            }
        }
        T_MARK(1)
        lds_barrier();
        T_MARK(2)
        const size_t tb = (size_t)(p0 + tile * ROWS) * 256;
        uint32_t* mk = STORE ? masks + (size_t)(p0 + tile * ROWS) * 16 : nullptr;         // 16 words per point: [layer 2][wave 4][row half NT][lane 64] per tile
        f32x16 acc[2][NT];
        f32x16 accc[2][H2 ? NT : 1];      // H2: the cross terms' accumulators
        RxBias bias = rx_load_bias(pf + RO_B6, wave, lane);
        WFrag3 nf = rx_gemm<RX_TH, NT, H2>(X, w_fw6, lane, acc, fr6, w_fw1, accc);          // F_color.6 on the weighted mean
        T_MARK(3)
        lds_barrier();
        T_MARK(2)
        rx_linear_epilogue<NT, H2>(X, acc, &bias, wave, lane);       // agg -> columns 0..255 (the dir-enc columns stay)
        T_MARK(4)
        lds_barrier();
        T_MARK(2)
        if (STORE) store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, agg + tb, 256, tid);              // kept for R.0's weight gradient
        T_MARK(5)
        bias = rx_load_bias(pf + RO_B1, wave, lane);
        fetch_slot(tile + (int)gridDim.x);
        nf = rx_gemm<RX_T1, NT, H2>(X, w_fw1, lane, acc, nf, w_fw2, accc);
        T_MARK(3)
        lds_barrier();
        T_MARK(2)
        rx_fwd_epilogue<STORE, NT, H2>(X, acc, bias, wave, lane, mk);
        T_MARK(4)
        lds_barrier();
        T_MARK(2)
        if (STORE) store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, act1 + tb, 256, tid);
        T_MARK(5)
        bias = rx_load_bias(pf + RO_B2, wave, lane);
        const RxBias w3q0 = rx_load_bias(pf + RO_W3, wave, lane), w3q1 = rx_load_bias(pf + RO_W3 + 256, wave, lane),
                     w3q2 = rx_load_bias(pf + RO_W3 + 512, wave, lane);      // the 3 x 256 last layer, this lane's quads: ahead of the GEMM
        rx_gemm<RX_TH, NT, H2>(X, w_fw2, lane, acc, nf, nullptr, accc);
        fetch_rows(tile + (int)gridDim.x);
        T_MARK(3)
        lds_barrier();
        T_MARK(2)
        {   // second activation (-> planes for the act2 store) and the 256 -> 3 layer from the registers: partial dot products per lane
            uint32_t bits[NT];
            float s3[NT][3];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bits[n] = 0u;
                s3[n][0] = s3[n][1] = s3[n][2] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
                    const f32x4 w3[3] = {w3q0.b[m][g], w3q1.b[m][g], w3q2.b[m][g]};
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 h, hs, out;
                        bias_scale4(acc[m][n], g, bias.b[m][g], h, hs);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            out[e] = lrelu_push(h[e], hs[e], bits[n]);
#pragma unroll
                            for (int c = 0; c < 3; ++c) s3[n][c] += w3[c][e] * out[e];
                        }
                        if (STORE) store_quad_xh<H2, RX_LDP>(X, 32 * n + j, f0, out);
                    }
                }
            if (STORE) {
#pragma unroll
                for (int n = 0; n < NT; ++n) mk[256 * NT + (NT * wave + n) * 64 + lane] = bits[n];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float t = s3[n][c] + __shfl_xor(s3[n][c], 32);
                    if (kg == 0) red[(wave * 64 + 32 * n + j) * 4 + c] = t;
                }
        }
        T_MARK(4)
        lds_barrier();
        T_MARK(2)
        if (STORE) store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, act2 + tb, 256, tid);
        T_MARK(5)
        if (tid < ROWS) {
            const int srow = s_row[tid];
            if (srow >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float t = ((red[tid * 4 + c] + red[(64 + tid) * 4 + c]) + (red[(128 + tid) * 4 + c] + red[(192 + tid) * 4 + c])) + pf[RO_B3 + c];
                    colors[(size_t)srow * 3 + c] = 1.f / (1.f + expf(-t));
                }
            }
        }
        T_MARK(6)
        lds_barrier();
        T_MARK(2)
    }
    T_FLUSH
}

template <bool STORE, bool H2 = false>
__global__ void __launch_bounds__(256, 1)
rhead_forward_x3_kernel(const float* __restrict__ agg3, const float* __restrict__ ray_dirs, const int32_t* __restrict__ point_slot,
                        const int32_t* __restrict__ n_points_dev, int max_points, int SR, const float* packed, float* __restrict__ colors,
                        float* __restrict__ agg, float* __restrict__ direnc, float* __restrict__ act1, float* __restrict__ act2,
                        uint32_t* __restrict__ masks) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * 64 * RX_LDP];
    __shared__ __attribute__((aligned(16))) float red[4 * 64 * 4];
    __shared__ int s_row[64];
    const int P = n_points_dev ? min(*n_points_dev, max_points) : max_points;
    // whole rounds of 64-point tiles first (every workgroup one tile per round); what is left — less than one round — as half-height tiles when
    // that puts it on twice the workgroups (55 k points on 256 workgroups: 3 rounds + 186 half tiles instead of a fourth round at 36 % occupancy)
    const int p_full = rx_full_rounds_points(P, (int)gridDim.x);
    if (p_full > 0)
        rhead_forward_x3_body<STORE, 2, H2>(X, red, s_row, agg3, ray_dirs, point_slot, 0, p_full, SR, packed, colors, agg, direnc, act1, act2, masks);
    if (P - p_full > 32 * (int)gridDim.x)
        rhead_forward_x3_body<STORE, 2, H2>(X, red, s_row, agg3, ray_dirs, point_slot, p_full, P, SR, packed, colors, agg, direnc, act1, act2, masks);
    else if (P > p_full)
        rhead_forward_x3_body<STORE, 1, H2>(X, red, s_row, agg3, ray_dirs, point_slot, p_full, P, SR, packed, colors, agg, direnc, act1, act2, masks);
}

// H2: as in the colour trunk's backward, every gradient ROW (= point) travels through the planes and the accumulators multiplied by its own power of two
// — a point's upstream gradient carries its compositing weight, 1 .. 1e-30 — chosen so that the row's largest |dL/d(pre-sigmoid)| lies in [1, 2): G2 =
// g3 W3 then stays within a few units and three more weight matrices have 2^14 of headroom below fp16's 65504 (the last product leaves from the fp32
// accumulators); every value that leaves the kernel is multiplied by the row's inverse factor (s_rinv).  The 3 x 256 layer's weight / bias gradient is
// formed from the UNSCALED g3.
template <int NT, bool H2 = false>
__device__ __forceinline__ void rhead_backward_x3_body(__bf16* X, float* s_g3, float* s_rinv, const float* __restrict__ g_colors, const float* __restrict__ colors,
                                                       const int32_t* __restrict__ point_slot, const int p0, const int P, const float* packed,
                                                       const float* __restrict__ act2, const uint32_t* __restrict__ masks, float* __restrict__ G1,
                                                       float* __restrict__ G2, float* __restrict__ g_agg, float* __restrict__ g_agg3,
                                                       float* __restrict__ g_w4 /* [3,256] */, float* __restrict__ g_b4 /* [3] */,
                                                       long long* __restrict__ g_w4_fixed, long long* __restrict__ g_b4_fixed) {
    constexpr int ROWS = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (P - p0 + ROWS - 1) / ROWS;             // points [p0, P), p0 a multiple of 64
    const float* packed0 = packed;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + (H2 ? RH_OFF : R_PACKED));
        gx3 w_bw2 = frag + RX_BW2 + wave * (RX_TH * 2 * 3 * 64) + lane;
        gx3 w_bwa = frag + RX_BWA + wave * (RX_TH * 2 * 3 * 64) + lane;
        gx3 w_bw6 = frag + RX_BW6 + wave * (RX_TH * 2 * 3 * 64) + lane;
        const size_t tb = (size_t)(p0 + tile * ROWS) * 256;
        const WFrag3 fr2 = load_wfrag3(w_bw2);               // in flight during the small last-layer stage
        if (tid < ROWS) {   // dL/d(pre-sigmoid) = g_c * c (1 - c)
            const int p = p0 + tile * ROWS + tid;
            float g[3] = {0.f, 0.f, 0.f};
            if (p < P) {
                const int srow = point_slot ? point_slot[p] : p;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float cc = colors[(size_t)srow * 3 + c];
                    g[c] = g_colors[(size_t)srow * 3 + c] * cc * (1.f - cc);
                }
            }
            float sc = 1.0f;
            if constexpr (H2) {
                const float rmax = fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fabsf(g[2]));
                int ex = 0;
                if (rmax > 0.f && rmax < 3.0e38f) (void)frexpf(rmax, &ex);          // rmax = m 2^ex, m in [0.5, 1)
                ex = max(ex, -100);                                                  // (the factor must stay finite: rows below 2^-100 keep 2^101)
                sc = ldexpf(1.0f, 1 - ex);                                           // row max -> [1, 2)
                s_rinv[tid] = ldexpf(1.0f, ex - 1);
            }
            *reinterpret_cast<f32x4*>(s_g3 + tid * 4) = f32x4{g[0], g[1], g[2], sc};   // [3] = the row's factor (H2), read by the G2 stage only
        }
        lds_barrier();
        const uint32_t* mk = masks + (size_t)(p0 + tile * ROWS) * 16;       // the forward's layout: 16 words per point, [layer 2][wave 4][row half NT][lane 64] per tile
        uint32_t bits[NT];                                                 // G2 stage's sign words, ahead of the dW3 stage
#pragma unroll
        for (int n = 0; n < NT; ++n) bits[n] = mk[256 * NT + (NT * wave + n) * 64 + lane];
        {   // dW3[c][col] += sum_rows g3[row][c] a2[row][col]; db3[c] += sum_rows g3[row][c]   (thread = column)
            float a0 = 0.f, a1 = 0.f, a2v = 0.f;
            const int rows_here = min(ROWS, P - p0 - tile * ROWS);
            for (int row = 0; row < rows_here; ++row) {
                const float a = act2[tb + row * 256 + tid];
                a0 += s_g3[row * 4] * a;
                a1 += s_g3[row * 4 + 1] * a;
                a2v += s_g3[row * 4 + 2] * a;
            }
            if (g_w4_fixed) {                 // order-independent fixed-point accumulation (common.h): bit-reproducible
                fixed_add(g_w4_fixed, tid, a0);
                fixed_add(g_w4_fixed, 256 + tid, a1);
                fixed_add(g_w4_fixed, 512 + tid, a2v);
            } else {
                atomicAdd(&g_w4[tid], a0);
                atomicAdd(&g_w4[256 + tid], a1);
                atomicAdd(&g_w4[512 + tid], a2v);
            }
            if (tid < 3) {
                float s = 0.f;
                for (int row = 0; row < ROWS; ++row) s += s_g3[row * 4 + tid];
                if (g_b4_fixed) fixed_add(g_b4_fixed, tid, s);
                else atomicAdd(&g_b4[tid], s);
            }
        }
        {   // G2 = (g3 W3) * lrelu'(h2), formed in the transposed accumulator arrangement -> planes
            f32x4 g3[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) g3[n] = *reinterpret_cast<const f32x4*>(s_g3 + (32 * n + j) * 4);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
                    f32x4 w3[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) w3[c] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(pf + RO_W3 + c * 256 + f0);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 out;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = g3[n][0] * w3[0][e] + g3[n][1] * w3[1][e] + g3[n][2] * w3[2][e];
                            if (H2) v *= g3[n][3];                               // the row's power of two (exact)
                            out[e] = lrelu_pop(v, v * 0.01f, bits[n]);
                        }
                        store_quad_xh<H2, RX_LDP>(X, 32 * n + j, f0, out);
                    }
                }
        }
        lds_barrier();
        store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, G2 + tb, 256, tid, s_rinv);
        f32x16 acc[2][NT];
        f32x16 accc[2][H2 ? NT : 1];
        uint32_t mw1[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) mw1[n] = mk[(NT * wave + n) * 64 + lane];
        WFrag3 nf = rx_gemm<RX_TH, NT, H2>(X, w_bw2, lane, acc, fr2, w_bwa, accc);
        lds_barrier();
        rx_bwd_epilogue<NT, H2>(X, acc, wave, lane, mw1);
        lds_barrier();
        store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, G1 + tb, 256, tid, s_rinv);
        nf = rx_gemm<RX_TH, NT, H2>(X, w_bwa, lane, acc, nf, w_bw6, accc);
        lds_barrier();
        rx_linear_epilogue<NT, H2>(X, acc, nullptr, wave, lane);     // g_agg: operand of F_color.6's weight gradient and of the last product
        lds_barrier();
        store_tile_from_planes_xh<H2, 32, RX_LDP, ROWS>(X, g_agg + tb, 256, tid, s_rinv);
        rx_gemm<RX_TH, NT, H2>(X, w_bw6, lane, acc, nf, nullptr, accc);                 // g_agg3 = g_agg W6
        lds_barrier();
        {   // -> fp32 tile in the (now dead) plane memory -> coalesced rows in HBM
            float* XF = reinterpret_cast<float*>(X);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        *reinterpret_cast<f32x4*>(XF + (32 * n + j) * LDA + 64 * wave + 32 * m + 8 * g + 4 * kg) =
                            f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
            lds_barrier();
#pragma unroll 4
            for (int u = 0; u < ROWS / 4; ++u) {           // ROWS rows of 256 floats, 16 bytes per thread
                const int e4 = tid + 256 * u, row = e4 >> 6, c4 = e4 & 63;
                f32x4 o = *reinterpret_cast<const f32x4*>(XF + row * LDA + 4 * c4);
                if (H2) o = o * s_rinv[row];
                *reinterpret_cast<f32x4*>(g_agg3 + tb + row * 256 + 4 * c4) = o;
            }
        }
        lds_barrier();
    }
}

template <bool H2 = false>
__global__ void __launch_bounds__(256, 1)
rhead_backward_x3_kernel(const float* __restrict__ g_colors, const float* __restrict__ colors, const int32_t* __restrict__ point_slot,
                         const int32_t* __restrict__ n_points_dev, int max_points, const float* packed, const float* __restrict__ act2,
                         const uint32_t* __restrict__ masks, float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ g_agg,
                         float* __restrict__ g_agg3, float* __restrict__ g_w4 /* [3,256] */, float* __restrict__ g_b4 /* [3] */,
                         long long* __restrict__ g_w4_fixed, long long* __restrict__ g_b4_fixed) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * 64 * RX_LDP];
    __shared__ __attribute__((aligned(16))) float s_g3[64 * 4];
    __shared__ float s_rinv[64];               // H2: the rows' inverse factors
    const int P = n_points_dev ? min(*n_points_dev, max_points) : max_points;
    // the forward's split (same P, same grid): whole rounds of 64-point tiles, the rest as half-height tiles if that is at most one per workgroup
    const int p_full = rx_full_rounds_points(P, (int)gridDim.x);
    if (p_full > 0)
        rhead_backward_x3_body<2, H2>(X, s_g3, s_rinv, g_colors, colors, point_slot, 0, p_full, packed, act2, masks, G1, G2, g_agg, g_agg3, g_w4, g_b4, g_w4_fixed, g_b4_fixed);
    if (P - p_full > 32 * (int)gridDim.x)
        rhead_backward_x3_body<2, H2>(X, s_g3, s_rinv, g_colors, colors, point_slot, p_full, P, packed, act2, masks, G1, G2, g_agg, g_agg3, g_w4, g_b4, g_w4_fixed, g_b4_fixed);
    else if (P > p_full)
        rhead_backward_x3_body<1, H2>(X, s_g3, s_rinv, g_colors, colors, point_slot, p_full, P, packed, act2, masks, G1, G2, g_agg, g_agg3, g_w4, g_b4, g_w4_fixed, g_b4_fixed);
}


// both images in one launch (the weights change every optimisation step: the packing is on the step's critical path)
__global__ void rhead_pack_kernel(RPackArgs a, float* __restrict__ out, float* __restrict__ zero_buf, long long zero_floats) {
    rhead_pack_all(a, out, zero_buf, zero_floats, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

}  // namespace

SPF_DEFINE_TIMING_ENTRY(spf_debug_timing_rhead)

extern "C" {

int64_t spf_rhead_packed_floats(void) { return R_PACKED_TOTAL; }

int spf_rhead_pack(const float* w6, const float* b6, const float* w0, const float* b0, const float* w2, const float* b2, const float* w4,
                   const float* b4, float* packed, float* zero_buf, int64_t zero_floats, void* stream) {
    if (!w6 || !b6 || !w0 || !b0 || !w2 || !b2 || !w4 || !b4 || !packed) return spf::fail(SPF_EINVAL, "spf_rhead_pack: null pointer");
    if (zero_floats < 0 || (zero_buf && ((uintptr_t)zero_buf & 15))) return spf::fail(SPF_EINVAL, "spf_rhead_pack: zero_buf must be 16-byte aligned, zero_floats >= 0");
    RPackArgs a{w6, b6, w0, b0, w2, b2, w4, b4};
    constexpr int NTH = R_PACK_THREADS;
    rhead_pack_kernel<<<spf::div_up(NTH, 256), 256, 0, (hipStream_t)stream>>>(a, packed, zero_buf, (long long)zero_floats);
    SPF_LAUNCH_CHECK("rhead_pack_kernel");
    return SPF_OK;
}

int spf_rhead_forward(const float* agg3, const float* ray_dirs, const int32_t* point_slot, const int32_t* n_points, int32_t max_points,
                      int32_t SR, const float* packed, float* colors, float* agg, float* direnc, float* act1, float* act2, uint32_t* masks,
                      int32_t arith, void* stream) {
    const bool h2 = arith == SPF_ARITH_H2;        // the 'split' family with three fp16 piece products (layouts, sign words, outputs unchanged)
    if (h2) arith = SPF_ARITH_SPLIT;
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_rhead_forward: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (max_points < 0 || SR < 1) return spf::fail(SPF_EINVAL, "spf_rhead_forward: bad sizes");
    if (max_points == 0) return SPF_OK;
    if (!agg3 || !ray_dirs || !packed || !colors) return spf::fail(SPF_EINVAL, "spf_rhead_forward: null pointer");
    const bool store = direnc != nullptr;
    if (store && (!agg || !act1 || !act2 || !masks)) return spf::fail(SPF_EINVAL, "spf_rhead_forward: training buffers must be given together");
    const int tiles = spf::div_up(max_points, 64);
    const int blocks = tiles < 512 ? tiles : 512;
    if (arith == SPF_ARITH_SPLIT) {
        const int t32 = spf::div_up(max_points, 32);
        const int b1 = t32 < 256 ? t32 : 256;           // one workgroup per CU; <= 32 points per workgroup: half-height tiles (chosen on the device)
        auto go = [&](auto kern) {
            kern<<<b1, 256, 0, (hipStream_t)stream>>>(agg3, ray_dirs, point_slot, n_points, max_points, SR, packed, colors, store ? agg : nullptr,
                                                     store ? direnc : nullptr, store ? act1 : nullptr, store ? act2 : nullptr, store ? masks : nullptr);
        };
        if (store) { if (h2) go(rhead_forward_x3_kernel<true, true>); else go(rhead_forward_x3_kernel<true, false>); }
        else { if (h2) go(rhead_forward_x3_kernel<false, true>); else go(rhead_forward_x3_kernel<false, false>); }
        SPF_LAUNCH_CHECK("rhead_forward_x3_kernel");
        return SPF_OK;
    }
    if (store)
        rhead_forward_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(agg3, ray_dirs, point_slot, n_points, max_points, SR, packed, colors, agg,
                                                                            direnc, act1, act2, masks);
    else
        rhead_forward_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(agg3, ray_dirs, point_slot, n_points, max_points, SR, packed, colors,
                                                                             nullptr, nullptr, nullptr, nullptr, nullptr);
    SPF_LAUNCH_CHECK("rhead_forward_kernel");
    return SPF_OK;
}

int spf_rhead_backward(const float* g_colors, const float* colors, const int32_t* point_slot, const int32_t* n_points, int32_t max_points,
                       const float* packed, const float* act2, const uint32_t* masks, float* G1, float* G2, float* g_agg, float* g_agg3,
                       float* g_b6, float* g_b0, float* g_b2, float* g_w4, float* g_b4, int64_t* g_w4_fixed, int64_t* g_b4_fixed, int32_t arith,
                       void* stream) {
    const bool h2 = arith == SPF_ARITH_H2;
    if (h2) arith = SPF_ARITH_SPLIT;
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_rhead_backward: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (max_points < 0) return spf::fail(SPF_EINVAL, "spf_rhead_backward: bad sizes");
    if ((g_w4_fixed || g_b4_fixed) && (arith != SPF_ARITH_SPLIT || !g_w4_fixed || !g_b4_fixed))
        return spf::fail(SPF_EINVAL, "spf_rhead_backward: the fixed-point accumulators come as a pair and need SPF_ARITH_SPLIT");
    if (max_points == 0) return SPF_OK;
    if (!g_colors || !colors || !packed || !act2 || !masks || !G1 || !G2 || !g_agg || !g_agg3 || !g_b6 || !g_b0 || !g_b2 || !g_w4 || !g_b4)
        return spf::fail(SPF_EINVAL, "spf_rhead_backward: null pointer");
    const int tiles = spf::div_up(max_points, 64);
    const int blocks = tiles < 512 ? tiles : 512;
    if (arith == SPF_ARITH_SPLIT) {    // g_b6 / g_b0 / g_b2 are not touched in this mode: spf_wgrad's dbias output provides them
        const int t32 = spf::div_up(max_points, 32);
        const int b1 = t32 < 256 ? t32 : 256;           // the forward's grid: both kernels then choose the same tile height
        auto go = [&](auto kern) {
            kern<<<b1, 256, 0, (hipStream_t)stream>>>(g_colors, colors, point_slot, n_points, max_points, packed, act2, masks, G1, G2, g_agg, g_agg3, g_w4,
                                                     g_b4, reinterpret_cast<long long*>(g_w4_fixed), reinterpret_cast<long long*>(g_b4_fixed));
        };
        if (h2) go(rhead_backward_x3_kernel<true>); else go(rhead_backward_x3_kernel<false>);
        SPF_LAUNCH_CHECK("rhead_backward_x3_kernel");
        return SPF_OK;
    }
    rhead_backward_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g_colors, colors, point_slot, n_points, max_points, packed, act2, masks, G1, G2,
                                                                   g_agg, g_agg3, g_b6, g_b0, g_b2, g_w4, g_b4);
    SPF_LAUNCH_CHECK("rhead_backward_kernel");
    return SPF_OK;
}

}  // extern "C"
