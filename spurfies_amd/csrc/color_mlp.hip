// Fused colour-feature path (K5 of DESIGN.md): positional encoding of the relative position +
// gather of the 64-d colour latent, the three activated layers of F_color (103 -> 256 -> 256 -> 256) on the fp32
// matrix cores and the RBF-weighted mean over a point's neighbours — forward, and the data-gradient chain of the
// backward with the colour-latent scatter-add.
//
// Replaces the F_color half of get_color, spurfies/model/pointneus_disent.py:325-336
// (positional encoding embedder.py:26-30, table gather utils.py:140-170, cuBLAS GEMMs,
// index_add_ of [pairs,256]) and autograd's backward through it.  F_color's LAST layer (F_color.6) is linear
// (pointneus_disent.py:76-85: no activation after it), so it commutes with the weighted mean:
//     sum_j wn_j (W6 a3_j + b6) = W6 (sum_j wn_j a3_j) + b6 sum_j wn_j,     sum_j wn_j = 1;
// this stage therefore outputs agg3[p] = sum_j wn_j a3_j and the 256x256 layer runs once per POINT (8x fewer rows)
// at the front of the head stage (rhead_mlp.hip), together with the `R` head (:338-346).
//
// Same tile engine as the geometry kernel (mlp_tile.h): tile = 8 points x 8 neighbour slots = 64
// rows, compact tile order (row = p*8 + j for the p-th valid point).
//
// Training mode stores each layer's input activation ([rows,104] and 3 x [rows,256]) so that
//   * the backward kernel recovers LeakyReLU' from the sign of the stored activation, and
//   * the weight gradients dW_l = G_l^T A_{l-1} are plain GEMMs over [rows,256] buffers (spf_wgrad, wgrad.hip)
//     (G_l = gradient w.r.t. layer l's pre-activation, stored by the backward kernel).
#include "mlp_tile.h"
#include "mlp_tile_x3.h"
#include "color_pack.h"

namespace {

using namespace spf;

// epilogue of a layer: + bias, (LeakyReLU), write to X.  Training mode also records the sign bits (one uint32 per
// lane and row half: bit n*16+r, the same lane/register position the backward kernel's accumulators have).
template <bool STORE, bool ACT>
__device__ __forceinline__ void c_fwd_epilogue(float* X, const f32x16 (&acc)[2][2], const float (&bv)[2], int wave, int lane,
                                               uint32_t* mask_g /* [4][2][64] or null */) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        uint32_t bits = 0u;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] + bv[n];
                if (ACT) {
                    const bool pos = v > 0.f;
                    bits |= (pos ? 1u : 0u) << (n * 16 + r);
                    v = pos ? v : v * 0.01f;
                }
                const int row = m * 32 + row_of(r, h);
                X[row * LDA + c0 + 32 * n] = v;
            }
        if (STORE && ACT && mask_g) mask_g[(wave * 2 + m) * 64 + lane] = bits;
    }
}

// Last activated layer: + bias, LeakyReLU (sign bits recorded), and the RBF-weighted sum over each point's rows taken straight
// from the accumulators — a lane holds two columns x 32 of the tile's 64 rows (the other 32 sit in lane ^ 32), in ascending row
// order, and a point's rows are consecutive, so each lane runs its own segmented sum and flushes one atomic add per (point,
// column) it has seen.  No activation tile is written, no barrier, no serial pass over 64 LDS rows per column.
template <bool STORE>
__device__ __forceinline__ void c_fwd_epilogue_reduce(const f32x16 (&acc)[2][2], const float (&bv)[2], int wave, int lane,
                                                      uint32_t* mask_g, const float* s_wp /* [64][2] = {weight, point id bits} */,
                                                      float* __restrict__ out) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
    int cur = -1;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        uint32_t bits = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keep the row bookkeeping reads from being hoisted en bloc (spills)
            const int row = m * 32 + row_of(r, h);
            const float2 wp = *reinterpret_cast<const float2*>(s_wp + 2 * row);
            const int p = __float_as_int(wp.y);
            const float w = wp.x;
            float v0 = acc[m][0][r] + bv[0], v1 = acc[m][1][r] + bv[1];
            const bool p0 = v0 > 0.f, p1 = v1 > 0.f;
            bits |= (p0 ? 1u : 0u) << r;
            bits |= (p1 ? 1u : 0u) << (16 + r);
            v0 = p0 ? v0 : v0 * 0.01f;
            v1 = p1 ? v1 : v1 * 0.01f;
            if (p != cur) {
                if (cur >= 0) {
                    atomicAdd(&out[(size_t)cur * 256 + c0], a0);
                    atomicAdd(&out[(size_t)cur * 256 + c0 + 32], a1);
                }
                cur = p;
                a0 = 0.f;
                a1 = 0.f;
            }
            a0 += w * v0;
            a1 += w * v1;
        }
        if (STORE && mask_g) mask_g[(wave * 2 + m) * 64 + lane] = bits;
    }
    if (cur >= 0) {
        atomicAdd(&out[(size_t)cur * 256 + c0], a0);
        atomicAdd(&out[(size_t)cur * 256 + c0 + 32], a1);
    }
}

// Layer-0 input of one tile: thread = (row, quarter): 16 colour-latent floats each + a share of the positional encoding of x_pi
// (internal columns [latent 64 | posenc 39 | pad]); thread 0 of a row also writes {weight, point id} for the weighted mean.
__device__ __forceinline__ void c_gather(float* X, float* s_wp, int ltid, int q, int idx, int srow, int p, const float* __restrict__ x,
                                         const float* __restrict__ pts, const float* __restrict__ feat_col, const float* __restrict__ wn) {
    const int row = ltid >> 2, q4 = ltid & 3;
    f32x4 f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx >= 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(feat_col + (size_t)idx * SPF_COL_DIM + q4 * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = src[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(X + row * LDA + q4 * 16 + 4 * u) = f[u];
    // positional encoding of x_pi, internal columns 64..102: the 18 (frequency, component) sin/cos pairs of a row are split over its
    // 4 threads; thread 0 also writes the raw offset, the pad column and the row's bookkeeping
    float d[3] = {0.f, 0.f, 0.f};
    if (idx >= 0) {
        d[0] = x[(size_t)srow * 3] - pts[(size_t)idx * 3];
        d[1] = x[(size_t)srow * 3 + 1] - pts[(size_t)idx * 3 + 1];
        d[2] = x[(size_t)srow * 3 + 2] - pts[(size_t)idx * 3 + 2];
    }
    float* e = X + row * LDA + 64;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
        const int j = q4 + 4 * jj;          // 0..17 -> (l, c)
        if (j < 3 * N_FREQ) {
            const int l = j / 3, c = j % 3;
            float sv = 0.f, cv = 0.f;
            if (idx >= 0) {
                const float a = d[c] * (float)(1 << l);
                sv = sinf(a);
                cv = cosf(a);
            }
            e[3 + 6 * l + c] = sv;
            e[6 + 6 * l + c] = cv;
        }
    }
    if (q4 == 0) {
        e[0] = d[0]; e[1] = d[1]; e[2] = d[2];
        e[39] = 0.f;                      // pad column 103
        s_wp[2 * row] = idx >= 0 ? wn[q] : 0.f;
        s_wp[2 * row + 1] = __int_as_float(idx >= 0 ? p : -1);
    }
}

template <bool STORE>
__global__ void __launch_bounds__(256, 2)
color_forward_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr, const float* __restrict__ wn,
                     const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point,
                     const int32_t* __restrict__ n_pairs_dev, int max_pairs, int k, const float* __restrict__ pts,
                     const float* __restrict__ feat_col, const float* packed, float* __restrict__ agg3, float* __restrict__ act0,
                     float* __restrict__ act1, float* __restrict__ act2, uint32_t* __restrict__ masks) {
#ifdef SPF_SOLO
    __shared__ __attribute__((aligned(16))) float smem[CL_TOTAL + 8000];     // experiment: one workgroup per CU
#else
    __shared__ __attribute__((aligned(16))) float smem[CL_TOTAL];
#endif
    float* X = smem + CL_X;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const float* packed0 = packed;
    // pair -> (point, slot row, neighbour) is a chain of three dependent reads: the NEXT tile's chain is walked one link per GEMM
    // of the current tile (n_*), so none of it is exposed at the top of a tile
    int n_p = -1, n_srow = 0, n_idx = -1;
    {
        const int q = blockIdx.x * 64 + (tid >> 2);
        if (blockIdx.x < ntiles && q < NP) {
            n_p = pair_point[q];
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_idx = nbr[(size_t)n_srow * k + (q - pair_off[n_p])];
        }
    }
    T_DECL

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp packed = launder(packed0);
        gf4p pk4 = reinterpret_cast<gf4p>(packed);
        const BFrag fr1 = load_bfrag(pk4 + (CO_FW1 / 4) + wave * (T_CIN * 128), lane);     // in flight during the gather
        T_MARK(15)
        const int qn = (tile + (int)gridDim.x) * 64 + (tid >> 2);                          // this thread's row in the block's next tile
        const bool has_next = tile + (int)gridDim.x < ntiles && qn < NP;
        c_gather(X, smem + CL_W, tid, tile * 64 + (tid >> 2), n_idx, n_srow, n_p, x, pts, feat_col, wn);
        T_MARK(0)
        __syncthreads();
        T_MARK(1)
        if (STORE) {  // layer-1 input [64][104] -> act0 (coalesced float4 copy out of LDS)
            float* dst = act0 + (size_t)tile * 64 * C_INP;
            for (int e4 = tid; e4 < 64 * (C_INP / 4); e4 += 256) {
                const int row = e4 / (C_INP / 4), c4 = e4 % (C_INP / 4);
                *reinterpret_cast<f32x4*>(dst + row * C_INP + 4 * c4) = *reinterpret_cast<const f32x4*>(X + row * LDA + 4 * c4);
            }
        }
        uint32_t* mk = STORE ? masks + (size_t)tile * 3 * 512 : nullptr;   // [layer 3][wave 4][m 2][lane 64]
        // bias values are requested before each GEMM, the next layer's first weight fragment inside it (mlp_tile.h)
        const int cb = wave * 64 + (lane & 31);
        gf4p wfw1 = pk4 + (CO_FW1 / 4) + wave * (T_CIN * 128);
        gf4p wfw2 = pk4 + (CO_FW2 / 4) + wave * (T_HID * 128);
        gf4p wfw3 = pk4 + (CO_FW3 / 4) + wave * (T_HID * 128);
        float bv[2] = {packed[CO_B1 + cb], packed[CO_B1 + cb + 32]};
        n_p = has_next ? pair_point[qn] : -1;
        f32x16 acc[2][2];
        zero_acc(acc);
        BFrag nf = gemm_rows64<T_CIN>(X, wfw1, lane, acc, fr1, wfw2);
        T_MARK(2)
        __syncthreads();
        T_MARK(3)
        c_fwd_epilogue<STORE, true>(X, acc, bv, wave, lane, mk);
        T_MARK(4)
        __syncthreads();
        T_MARK(5)
        bv[0] = packed[CO_B2 + cb]; bv[1] = packed[CO_B2 + cb + 32];
        int n_off = 0;
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        zero_acc(acc);
        nf = gemm_rows64<T_HID>(X, wfw2, lane, acc, nf, wfw3, side_tile(STORE ? act1 + (size_t)tile * 64 * 256 : nullptr, tid));
        T_MARK(6)
        __syncthreads();
        T_MARK(7)
        c_fwd_epilogue<STORE, true>(X, acc, bv, wave, lane, STORE ? mk + 512 : nullptr);
        T_MARK(8)
        __syncthreads();
        T_MARK(9)
        bv[0] = packed[CO_B3 + cb]; bv[1] = packed[CO_B3 + cb + 32];
        n_idx = n_p >= 0 ? nbr[(size_t)n_srow * k + (qn - n_off)] : -1;
        zero_acc(acc);
        gemm_rows64<T_HID>(X, wfw3, lane, acc, nf, nullptr, side_tile(STORE ? act2 + (size_t)tile * 64 * 256 : nullptr, tid));
        T_MARK(10)
        // agg3[p] = sum_j wn_j a3_j straight from the accumulators; the linear F_color.6 follows per point (rhead_mlp.hip)
        c_fwd_epilogue_reduce<STORE>(acc, bv, wave, lane, STORE ? mk + 1024 : nullptr, smem + CL_W, agg3);
        T_MARK(13)
        __syncthreads();       // X, s_w, s_p are rewritten by the next tile's gather
        T_MARK(14)
    }
    T_FLUSH
}

// backward epilogue: G_l = g_a * lrelu'(h_l) with the sign bits the forward recorded; write X and G_l (operand of the
// wgrad GEMM) and add this tile's column sums to the bias gradient (256 floats per layer per tile, 128-B atomics)
__device__ __forceinline__ void c_bwd_epilogue(float* X, const f32x16 (&acc)[2][2], int wave, int lane,
                                               const uint32_t (&mbits)[2], float* __restrict__ g_bias) {
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
                X[row * LDA + c0 + 32 * n] = v;
                cs[n] += v;
            }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float t = cs[n] + __shfl_xor(cs[n], 32);
        if (h == 0) atomicAdd(&g_bias[c0 + 32 * n], t);
    }
}

// same, for a gradient tile that is already in LDS (no GEMM in front): every lane owns the (row, column) positions its
// accumulators would have, so the in-place update needs no further synchronisation
__device__ __forceinline__ void c_bwd_mask_inplace(float* X, int wave, int lane, const uint32_t (&mbits)[2], float* __restrict__ g_bias) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
    float cs[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const uint32_t bits = mbits[m];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* px = X + (m * 32 + row_of(r, h)) * LDA + c0 + 32 * n;
                float v = *px;
                v = ((bits >> (n * 16 + r)) & 1u) ? v : v * 0.01f;
                *px = v;
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
color_backward_kernel(const float* __restrict__ g_agg3, const int32_t* __restrict__ nbr, const float* __restrict__ wn,
                      const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point,
                      const int32_t* __restrict__ n_pairs_dev, int max_pairs, int k, const float* packed,
                      const uint32_t* __restrict__ masks, float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ G3,
                      float* __restrict__ g_b0, float* __restrict__ g_b2, float* __restrict__ g_b4 /* bias gradients [256] of layers 0, 2, 4 */,
                      float* __restrict__ g_feat_col) {
    __shared__ __attribute__((aligned(16))) float smem[CL_TOTAL];
    float* X = smem + CL_X;
    int* s_idx = reinterpret_cast<int*>(smem + CL_W);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const float* packed0 = packed;
    // the next tile's (point, weight) are fetched during the current tile; the neighbour index — only needed by the final
    // latent scatter — is resolved link by link behind the current tile's GEMMs
    int n_p = -1;
    float n_w = 0.f;
    {
        const int q = blockIdx.x * 64 + (tid >> 2);
        if (blockIdx.x < ntiles && q < NP) {
            n_p = pair_point[q];
            n_w = wn[q];
        }
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp packed = launder(packed0);
        gf4p pk4 = reinterpret_cast<gf4p>(packed);
        const size_t tbase = (size_t)tile * 64 * 256;
        const uint32_t* mk = masks + (size_t)tile * 3 * 512;
        const BFrag fr3 = load_bfrag(pk4 + (CO_BW3 / 4) + wave * (T_HID * 128), lane);    // in flight during the gather
        const uint32_t mb3[2] = {mk[1024 + (wave * 2) * 64 + lane], mk[1024 + (wave * 2 + 1) * 64 + lane]};
        // ---- g_a3[row] = wn[row] * g_agg3[p]  (agg3 = sum_j wn_j a3_j) -------------------------------
        const int qrow = tile * 64 + (tid >> 2);
        const int p_cur = n_p;                       // fetched during the previous tile (or before the loop)
        {
            const int row = tid >> 2, q4 = tid & 3;
            const float w = n_w;
            const int p = p_cur < 0 ? 0 : p_cur;
            const f32x4* ga = reinterpret_cast<const f32x4*>(g_agg3 + (size_t)p * 256);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c4 = q4 + 4 * u;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (w != 0.f) {
                    v = ga[c4];
                    v[0] *= w; v[1] *= w; v[2] *= w; v[3] *= w;
                }
                *reinterpret_cast<f32x4*>(X + row * LDA + 4 * c4) = v;
            }
        }
        __syncthreads();
        f32x16 acc[2][2];
        gf4p wbw3 = pk4 + (CO_BW3 / 4) + wave * (T_HID * 128);
        gf4p wbw2 = pk4 + (CO_BW2 / 4) + wave * (T_HID * 128);
        c_bwd_mask_inplace(X, wave, lane, mb3, g_b4);     // G3 = g_a3 * lrelu'(h3), in place
        __syncthreads();
        int c_srow = 0, c_off = 0;
        if (p_cur >= 0) {
            c_srow = point_slot ? point_slot[p_cur] : p_cur;
            c_off = pair_off[p_cur];
        }
        {
            const int qn = (tile + (int)gridDim.x) * 64 + (tid >> 2);
            const bool has_next = tile + (int)gridDim.x < ntiles && qn < NP;
            n_p = has_next ? pair_point[qn] : -1;
            n_w = has_next ? wn[qn] : 0.f;
        }
        uint32_t mb[2] = {mk[512 + (wave * 2) * 64 + lane], mk[512 + (wave * 2 + 1) * 64 + lane]};   // sign bits: requested before the GEMM
        zero_acc(acc);
        BFrag nf = gemm_rows64<T_HID>(X, wbw3, lane, acc, fr3, wbw2, side_tile(G3 + tbase, tid));
        __syncthreads();
        c_bwd_epilogue(X, acc, wave, lane, mb, g_b2);
        __syncthreads();
        if ((tid & 3) == 0) s_idx[tid >> 2] = p_cur >= 0 ? nbr[(size_t)c_srow * k + (qrow - c_off)] : -1;   // read after two more barriers
        mb[0] = mk[(wave * 2) * 64 + lane]; mb[1] = mk[(wave * 2 + 1) * 64 + lane];
        zero_acc(acc);
        gemm_rows64<T_HID>(X, wbw2, lane, acc, nf, nullptr, side_tile(G2 + tbase, tid));
        __syncthreads();
        c_bwd_epilogue(X, acc, wave, lane, mb, g_b0);
        __syncthreads();
        store_tile_256(X, G1 + tbase, tid);
        // ---- d/d latent = G1 * W0[:, 39:103]; wave = (row half mt, latent half nt); scatter-add ------
        {
            const int mt = wave >> 1, nt = wave & 1, i = lane & 31, h = lane >> 5;
            f32x16 aj;
#pragma unroll
            for (int r = 0; r < 16; ++r) aj[r] = 0.f;
            const float* ap = X + (mt * 32 + i) * LDA + 4 * h;
            gf4p bp = pk4 + (CO_BWL / 4) + nt * (T_HID * 64) + lane;
#pragma unroll 4
            for (int t = 0; t < T_HID; ++t) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * t);
                const f32x4 b = bp[t * 64];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) aj = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj], b[jj], aj, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + row_of(r, h);
                const int idx = s_idx[row];
                if (idx >= 0) atomicAdd(&g_feat_col[(size_t)idx * SPF_COL_DIM + 32 * nt + i], aj[r]);
            }
        }
        __syncthreads();
    }
}


// ==============================================================================================================================
// The same two kernels on the bf16 matrix pipe with fp32-CLASS products (six exact bf16 piece products, <= 2 ulp per fp32 product) from three bf16 pieces per operand (arith = SPF_ARITH_SPLIT; rounds 1 - 5's default)
// or, since round 6, three exact fp16 piece products from two fp16 pieces per operand (arith = SPF_ARITH_H2: the template flag H2 below; what the Python layer
// passes by default); arith = SPF_ARITH_F32 selects the fp32-MFMA kernels above.  Engine and arithmetic argument: mlp_tile_x3.h / geo_mlp.hip.
//   * layers 0 and 2 run as transposed products (a lane owns 4 consecutive features of one row, the epilogue rewrites the bf16
//     planes with 8-byte stores); layer 4 — whose output only feeds the RBF-weighted mean — runs non-transposed, so that a lane
//     owns a column strip and takes the segmented weighted sum straight from its accumulators, as the fp32 kernel does;
//   * LeakyReLU sign bits, masks[tile][layer][512 words]: layers 0 and 2 lane-major ([wave][n][lane], the forward and backward
//     transposed epilogues share the lane <-> element map and push / pop the words in the same order); layer 4 ROW-major
//     ([row][8 words], bit = feature & 31: a row word per accumulator register from the wave ballot, read back by the
//     backward's first stage one row quarter per thread);
//   * what the weight-gradient GEMM reads: act1 / act2 (forward) and G2 / G1 (backward) leave the transposed epilogues straight
//     from the registers as K-major blocks [256 features][16 rows] (SPF_WGRAD_*_TILES; exactly the values the planes hold, since
//     p1 + p2 + p3 reproduces the fp32 value); G3 likewise from the backward's first stage (thread = (row = lane, feature quarter = wave));
//     act0 — produced by the (row, quarter) gather stage — is written as fp32 rows rebuilt from the planes; bias gradients come from the weight-gradient GEMM (column sums of G, spf_wgrad).
// ==============================================================================================================================
// (layout constants CX_* and the fragment packing: color_pack.h)


// this lane's bias values of a layer (features 64 wave + 32 m + 8 g + 4 kg ..+3), requested ahead of the layer's GEMM
struct CxBias {
    f32x4 b[2][4];
};
__device__ __forceinline__ CxBias cx_load_bias(gfp bias, int wave, int lane) {
    CxBias r;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            r.b[m][g] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(bias + 64 * wave + 32 * m + 8 * g + 4 * (lane >> 5));
    return r;
}

// transposed forward epilogue: a = lrelu(acc + b) -> planes; the lane's sign bits are pushed into two words (n = 0, 1; order
// (m, g, e)) and stored lane-major (STORE): masks_l[(2 wave + n) * 64 + lane] — the backward's transposed epilogue has the same
// lane <-> element map and pops them in the same order
// STORE also writes the activations to `tile_out`, the tile's K-MAJOR image (four blocks of [256 features][16 rows] fp32) the
// weight-gradient GEMM reads (SPF_WGRAD_*_TILES): straight from the registers — for each of a lane's four features 16 lanes cover
// 64 contiguous bytes — instead of a second pass that rebuilds fp32 rows from the planes.
template <bool STORE, bool H2 = false>
__device__ __forceinline__ void cx_fwd_epilogue(__bf16* X, const f32x16 (&acc)[2][2], const CxBias& bias, int wave, int lane, uint32_t* masks_l,
                                                float* __restrict__ tile_out) {
    const int j = lane & 31, kg = lane >> 5;
    uint32_t bits[2] = {0u, 0u};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 h, hs;
                bias_scale4(acc[m][n], g, bias.b[m][g], h, hs);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = lrelu_push(h[e], hs[e], bits[n]);
                store_quad_xh<H2, X3_LDP>(X, 32 * n + j, f0, out);
                if (STORE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) tile_out[(2 * n + (j >> 4)) * 4096 + (f0 + e) * 16 + (j & 15)] = out[e];
                }
            }
        }
    if (STORE) {
        masks_l[(2 * wave) * 64 + lane] = bits[0];
        masks_l[(2 * wave + 1) * 64 + lane] = bits[1];
    }
}

// H2 backward: gradients are small and their magnitude differs from ROW to row by many orders (a pair's row carries its RBF weight:
// exp(-(45 d)^2) spans 1 .. 1e-20; the rgb loss adds 1 / (3 R)), far outside fp16's range.  The chain is linear and rows (pairs) are independent,
// so every row travels through the planes and the accumulators multiplied by its OWN power of two (block floating point: the row's largest
// |G3| entry becomes 128 .. 256, leaving 2^8 of headroom for growth through the two weight matrices and 2^22 below it in fp16's normal range),
// and every value that leaves the kernel is multiplied by the row's inverse factor (both exact).  s_rinv[row] holds that inverse.

// transposed backward epilogue: g_h = g_a * lrelu'(h), popping the lane's two sign words -> planes
template <bool H2 = false>
__device__ __forceinline__ void cx_bwd_epilogue(__bf16* X, const f32x16 (&acc)[2][2], int wave, int lane, const uint32_t (&mw)[2],
                                                float* __restrict__ tile_out, const float* s_rinv = nullptr) {
    const int j = lane & 31, kg = lane >> 5;
    const float rinv[2] = {H2 ? s_rinv[j] : 1.0f, H2 ? s_rinv[32 + j] : 1.0f};
    uint32_t bits[2] = {mw[0], mw[1]};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 v, vs;
                scale4(acc[m][n], g, v, vs);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = lrelu_pop(v[e], vs[e], bits[n]);
                store_quad_xh<H2, X3_LDP>(X, 32 * n + j, f0, out);
#pragma unroll
                for (int e = 0; e < 4; ++e) tile_out[(f0 + e) * 64 + 32 * n + j] = H2 ? out[e] * rinv[n] : out[e];      // K-major 64-row tile (SPF_WGRAD_G_TILES64): full 128-byte lines
            }
        }
}

// one pair row's gather operands: a quarter of the colour latent, the offset x - p_i, the RBF weight and the point index
struct CxRow {
    f32x4 f[4];
    float d[3];
    float w;
    int p, idx;
};
__device__ __forceinline__ CxRow cx_fetch_row(int idx, int srow, int p, int q, int q4, const float* __restrict__ x, const float* __restrict__ pts,
                                              const float* __restrict__ feat_col, const float* __restrict__ wn) {
    CxRow r;
    r.idx = idx;
    r.p = p;
    r.w = 0.f;
    r.d[0] = r.d[1] = r.d[2] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) r.f[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx >= 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(feat_col + (size_t)idx * SPF_COL_DIM + q4 * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) r.f[u] = src[u];
        r.d[0] = x[(size_t)srow * 3] - pts[(size_t)idx * 3];
        r.d[1] = x[(size_t)srow * 3 + 1] - pts[(size_t)idx * 3 + 1];
        r.d[2] = x[(size_t)srow * 3 + 2] - pts[(size_t)idx * 3 + 2];
        r.w = wn[q];
    }
    return r;
}

#ifdef SPF_NO_AGG_ATOMICS      // timing-only ablation (wrong results): what the weighted mean's atomics cost
#define SPF_AGG_ATOMIC(p, v) ((void)(p), (void)(v))
#else
#define SPF_AGG_ATOMIC(p, v) atomicAdd(p, v)
#endif
constexpr int CX_LDS_BF16 = 3 * X3_PLANE;
constexpr int CX_LDL = 68;

// a layer's GEMM on the colour kernels' engine: bf16 x 3 (one accumulator) or H2 (main + cross accumulators, combined here)
template <int T, bool SWAP, bool H2, int NC>
__device__ __forceinline__ WFrag3 cx_gemm(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][2], const WFrag3& first, gx3 next_wp, f32x16 (&accc)[2][NC]) {
    if constexpr (H2) {
        static_assert(NC == 2, "H2: one cross accumulator per main accumulator");
        const WFrag3 nf = gemm_x3<T, SWAP, X3_LDP, 2, 2, true>(X, wp, lane, acc, first, next_wp, accc);
        h2_combine<2>(acc, accc);
        return nf;
    } else {
        return gemm_x3<T, SWAP>(X, wp, lane, acc, first, next_wp);
    }
}

template <bool STORE, bool H2 = false>
__global__ void __launch_bounds__(256, 1)
color_forward_x3_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr, const float* __restrict__ wn,
                        const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point,
                        const int32_t* __restrict__ n_pairs_dev, int max_pairs, int k, const float* __restrict__ pts,
                        const float* __restrict__ feat_col, const float* packed, float* __restrict__ agg3, float* __restrict__ act0,
                        float* __restrict__ act1, float* __restrict__ act2, uint32_t* __restrict__ masks, long long* __restrict__ agg3_fixed) {
    __shared__ __attribute__((aligned(16))) __bf16 X[CX_LDS_BF16];
    __shared__ __attribute__((aligned(8))) float s_wp[128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const float* packed0 = packed;
    T_DECL
    CxRow cur;
    {
        const int q = blockIdx.x * 64 + (tid >> 2);
        int p = -1, srow = 0, idx = -1;
        if (q < NP) {
            p = pair_point[q];
            srow = point_slot ? point_slot[p] : p;
            idx = nbr[(size_t)srow * k + (q - pair_off[p])];
        }
        cur = cx_fetch_row(idx, srow, p, q, tid & 3, x, pts, feat_col, wn);
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + (H2 ? CH_OFF : C_PACKED));
        gx3 w_fw1 = frag + CX_FW1 + wave * (CX_T1 * 2 * 3 * 64) + lane;
        gx3 w_fw2 = frag + CX_FW2 + wave * (CX_TH * 2 * 3 * 64) + lane;
        gx3 w_fw3 = frag + CX_FW3 + wave * (CX_TH * 2 * 3 * 64) + lane;
        const WFrag3 fr1 = load_wfrag3(w_fw1);
        T_MARK(15)
        // ---- gather: thread = (row, quarter): 16 latent floats + a share of the positional encoding, pieces into the planes; the
        //      operands were requested during the previous tile (lookup chain pair -> point -> slot -> neighbour -> latent row)
        {
            const int row = tid >> 2, q4 = tid & 3;
            const int idx = cur.idx;
            // training: the layer-0 input row [64 latent | 3 offset | 36 sin / cos | 0] also goes to act0 (fp32 rows, what the weight-gradient
            // GEMM of F_color.0 reads) straight from these registers — the values the planes hold, without a pass that rebuilds them
            float* a0 = STORE ? act0 + ((size_t)tile * 64 + row) * C_INP : nullptr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v[4] = {cur.f[u][0], cur.f[u][1], cur.f[u][2], cur.f[u][3]};
                store_quad_xh<H2, X3_LDP>(X, row, q4 * 16 + 4 * u, v);
                if (STORE) *reinterpret_cast<f32x4*>(a0 + q4 * 16 + 4 * u) = cur.f[u];
            }
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int jx = q4 + 4 * jj;          // 0..17 -> (l, c)
                if (jx < 3 * N_FREQ) {
                    const int l = jx / 3, c = jx % 3;
                    float sv = 0.f, cv = 0.f;
                    if (idx >= 0) {
                        const float a = cur.d[c] * (float)(1 << l);
                        sv = sinf(a);
                        cv = cosf(a);
                    }
                    store_one_xh<H2, X3_LDP>(X, row, 64 + 3 + 6 * l + c, sv);
                    store_one_xh<H2, X3_LDP>(X, row, 64 + 6 + 6 * l + c, cv);
                    if (STORE) {
                        a0[64 + 3 + 6 * l + c] = sv;
                        a0[64 + 6 + 6 * l + c] = cv;
                    }
                }
            }
            if (q4 == 0) {
                store_one_xh<H2, X3_LDP>(X, row, 64, cur.d[0]);
                store_one_xh<H2, X3_LDP>(X, row, 65, cur.d[1]);
                store_one_xh<H2, X3_LDP>(X, row, 66, cur.d[2]);
                store_one_xh<H2, X3_LDP>(X, row, 103, 0.f);          // pad column of the 104-wide internal layout
                if (STORE) {
                    a0[64] = cur.d[0];
                    a0[65] = cur.d[1];
                    a0[66] = cur.d[2];
                    a0[103] = 0.f;
                }
                const float z[4] = {0.f, 0.f, 0.f, 0.f};
                store_quad_xh<H2, X3_LDP>(X, row, 104, z);           // K padded to 112
                store_quad_xh<H2, X3_LDP>(X, row, 108, z);
                s_wp[2 * row] = cur.w;
                s_wp[2 * row + 1] = __int_as_float(idx >= 0 ? cur.p : -1);
            }
        }
        const int qn = (tile + (int)gridDim.x) * 64 + (tid >> 2);      // this thread's row in the workgroup's next tile
        int n_p = qn < NP ? pair_point[qn] : -1, n_srow = 0, n_off = 0, n_idx = -1;
        T_MARK(0)
        lds_barrier();
        T_MARK(1)
        uint32_t* mk = STORE ? masks + (size_t)tile * 3 * 512 : nullptr;                             // [layer 3][row 64][8 words]
        f32x16 acc[2][2];
        f32x16 accc[2][H2 ? 2 : 1];        // H2: the cross terms' accumulators
        CxBias bias = cx_load_bias(pf + CO_B1, wave, lane);
        zero_acc(acc);
        WFrag3 nf = cx_gemm<CX_T1, false, H2>(X, w_fw1, lane, acc, fr1, w_fw2, accc);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        cx_fwd_epilogue<STORE, H2>(X, acc, bias, wave, lane, mk, STORE ? act1 + (size_t)tile * 64 * 256 : nullptr);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        bias = cx_load_bias(pf + CO_B2, wave, lane);
        zero_acc(acc);
        nf = cx_gemm<CX_TH, false, H2>(X, w_fw2, lane, acc, nf, w_fw3, accc);
        T_MARK(6)
        lds_barrier();
        T_MARK(7)
        cx_fwd_epilogue<STORE, H2>(X, acc, bias, wave, lane, STORE ? mk + 512 : nullptr, STORE ? act2 + (size_t)tile * 64 * 256 : nullptr);
        T_MARK(8)
        lds_barrier();
        T_MARK(9)
        if (n_p >= 0) n_idx = nbr[(size_t)n_srow * k + (qn - n_off)];
        const float bv[2] = {pf[CO_B3 + 64 * wave + (lane & 31)], pf[CO_B3 + 64 * wave + (lane & 31) + 32]};     // layer-4 biases, ahead of the GEMM
        // ---- layer 4, NON-transposed: acc[m][n] = rows 32m.., features 64w + 32n..; lane (feature j, k-half kg) ------------------
        zero_acc(acc);
        cx_gemm<CX_TH, true, H2>(X, w_fw3, lane, acc, nf, nullptr, accc);
        T_MARK(10)
        cur = cx_fetch_row(n_idx, n_srow, n_p, qn, tid & 3, x, pts, feat_col, wn);      // lands during the epilogue
        // ---- + bias, LeakyReLU, sign words by ballot, RBF-weighted segmented sum from the accumulators -> atomics ---------------
        {
            const int j = lane & 31, kg = lane >> 5;
            const int c0 = 64 * wave + j;
            int cur = -1;
            float a0 = 0.f, a1 = 0.f;
            uint32_t mw0 = 0u, mw1 = 0u;      // lane l collects the two sign words of row l (features 64 wave + 0..31 and + 32..63)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * m + 8 * (r >> 2) + 4 * kg + (r & 3);
                    const float2 wp = *reinterpret_cast<const float2*>(s_wp + 2 * row);
                    const int p = __float_as_int(wp.y);
                    float v0 = acc[m][0][r] + bv[0], v1 = acc[m][1][r] + bv[1];
                    const bool p0 = v0 > 0.f, p1 = v1 > 0.f;
                    if (STORE) {     // ballot: low 32 lanes = row (kg = 0), high 32 = row + 4 (kg = 1); bit = feature & 31
                        const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1);
                        const int rlo = 32 * m + 8 * (r >> 2) + (r & 3);          // compile-time lane selects
                        // (a VALU-written SGPR needs 4 wait states before v_writelane reads it; the hazard recogniser does not see
                        // into inline asm)
                        asm("s_nop 3\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %1, %3, %6\n\tv_writelane_b32 %0, %4, %7\n\t"
                            "v_writelane_b32 %1, %5, %7"
                            : "+v"(mw0), "+v"(mw1)
                            : "s"((uint32_t)b0), "s"((uint32_t)b1), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)(b1 >> 32)), "n"(rlo), "n"(rlo + 4));
                    }
                    v0 = p0 ? v0 : v0 * 0.01f;
                    v1 = p1 ? v1 : v1 * 0.01f;
                    if (p != cur) {
                        if (cur >= 0) {
                            if (agg3_fixed) {          // order-independent accumulation (common.h): up to four partial sums meet per entry
                                fixed_add(agg3_fixed, (size_t)cur * 256 + c0, a0);
                                fixed_add(agg3_fixed, (size_t)cur * 256 + c0 + 32, a1);
                            } else {
                                SPF_AGG_ATOMIC(&agg3[(size_t)cur * 256 + c0], a0);
                                SPF_AGG_ATOMIC(&agg3[(size_t)cur * 256 + c0 + 32], a1);
                            }
                        }
                        cur = p;
                        a0 = 0.f;
                        a1 = 0.f;
                    }
                    a0 += wp.x * v0;
                    a1 += wp.x * v1;
                }
            if (cur >= 0) {
                if (agg3_fixed) {
                    fixed_add(agg3_fixed, (size_t)cur * 256 + c0, a0);
                    fixed_add(agg3_fixed, (size_t)cur * 256 + c0 + 32, a1);
                } else {
                    SPF_AGG_ATOMIC(&agg3[(size_t)cur * 256 + c0], a0);
                    SPF_AGG_ATOMIC(&agg3[(size_t)cur * 256 + c0 + 32], a1);
                }
            }
            if (STORE) *reinterpret_cast<u32x2*>(mk + 1024 + lane * 8 + 2 * wave) = u32x2{mw0, mw1};      // [row][8 words]
        }
        T_MARK(13)
        lds_barrier();       // planes and s_wp are rewritten by the next tile's gather
        T_MARK(14)
    }
    T_FLUSH
}

// one pair row's operands of the backward's first stage: a quarter of g_agg3[p], the RBF weight, the neighbour index and the
// two layer-3 sign words of the quarter
struct CxGrow {
    f32x4 ga[16];
    float w;
    int idx;
    uint32_t m0, m1;
};
__device__ __forceinline__ CxGrow cx_fetch_grow(const float* __restrict__ g_agg3, const uint32_t* __restrict__ mk, int p, int idx, float w, int row,
                                                int q4) {
    CxGrow r;
    r.w = w;
    r.idx = idx;
    r.m0 = mk[1024 + row * 8 + 2 * q4];
    r.m1 = mk[1024 + row * 8 + 2 * q4 + 1];
    const f32x4* ga = reinterpret_cast<const f32x4*>(g_agg3 + (size_t)(p < 0 ? 0 : p) * 256 + 64 * q4);
#pragma unroll
    for (int u = 0; u < 16; ++u) r.ga[u] = ga[u];
    return r;
}

template <bool H2>
__global__ void __launch_bounds__(256, 1)
color_backward_x3_kernel(const float* __restrict__ g_agg3, const int32_t* __restrict__ nbr, const float* __restrict__ wn,
                         const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point,
                         const int32_t* __restrict__ n_pairs_dev, int max_pairs, int k, const float* packed,
                         const uint32_t* __restrict__ masks, float* __restrict__ G1, float* __restrict__ G2, float* __restrict__ G3,
                         float* __restrict__ g_feat_col, long long* __restrict__ g_fixed) {
    __shared__ __attribute__((aligned(16))) __bf16 X[CX_LDS_BF16];
    __shared__ __attribute__((aligned(16))) int s_idx[64];
    __shared__ __attribute__((aligned(16))) float L[64 * CX_LDL];      // the tile's latent gradients, [row][64 (+4)]
    __shared__ __attribute__((aligned(16))) int s_lead[64];            // neighbour index of the rows that lead a group of equal indices, else -1
    __shared__ float s_amax[4][64];                                    // H2: per (feature quarter, row) largest |G3| entry, then ...
    __shared__ float s_rinv[64];                                       // ... the row's inverse power-of-two factor
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const float* packed0 = packed;
    T_DECL
    // first stage: thread = (pair row = lane, quarter of the 256 features = wave): a feature's 64 rows are then 64 consecutive
    // lanes, i.e. the K-major G3 tile leaves the registers in fully used 128-byte lines
    CxGrow cur;
    {
        const int q = blockIdx.x * 64 + lane;
        int p = -1, idx = -1;
        float w = 0.f;
        if (q < NP) {
            p = pair_point[q];
            const int srow = point_slot ? point_slot[p] : p;
            idx = nbr[(size_t)srow * k + (q - pair_off[p])];
            w = wn[q];
        }
        cur = cx_fetch_grow(g_agg3, masks + (size_t)blockIdx.x * 3 * 512, p, idx, w, lane, wave);
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + (H2 ? CH_OFF : C_PACKED));
        gx3 w_bw3 = frag + CX_BW3 + wave * (CX_TH * 2 * 3 * 64) + lane;
        gx3 w_bw2 = frag + CX_BW2 + wave * (CX_TH * 2 * 3 * 64) + lane;
        const WFrag3 fr3 = load_wfrag3(w_bw3);
        const size_t tbase = (size_t)tile * 64 * 256;
        const uint32_t* mk = masks + (size_t)tile * 3 * 512;
        // ---- G3[row] = wn[row] g_agg3[p] * lrelu'(h3): thread = (row, quarter of the 256 features) -> planes; the operands were
        //      requested during the previous tile
        const int row0 = tid >> 2, q40 = tid & 3;              // (row, quarter) of the latent-gradient stages below
        {
            if (wave == 0) s_idx[lane] = cur.idx;
            float* g3t = G3 + tbase + (size_t)(64 * wave) * 64 + lane;       // K-major 64-row tile (SPF_WGRAD_G_TILES64): feature f, row r at f * 64 + r
            float oall[16][4];
            float amax = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t word = u < 8 ? cur.m0 : cur.m1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pos = (word >> ((4 * u + e) & 31)) & 1u;
                    const float t = cur.w != 0.f ? cur.ga[u][e] * cur.w : 0.f;
                    const float o = pos ? t : t * 0.01f;
                    oall[u][e] = o;
                    g3t[(4 * u + e) * 64] = o;
                    amax = fmaxf(amax, fabsf(o));
                }
                if (!H2) store_quad_x3(X, lane, 64 * wave + 4 * u, oall[u]);
            }
            if (H2) {      // the row's power-of-two factor from the four feature quarters' maxima (one LDS round trip + a barrier)
                s_amax[wave][lane] = amax;
                lds_barrier();
                const float rmax = fmaxf(fmaxf(s_amax[0][lane], s_amax[1][lane]), fmaxf(s_amax[2][lane], s_amax[3][lane]));
                int ex = 0;
                if (rmax > 0.f && rmax < 3.0e38f) (void)frexpf(rmax, &ex);          // rmax = m 2^ex, m in [0.5, 1)
                ex = max(ex, -100);      // (rows below 2^-100 — an RBF weight of 1e-20 times a vanishing upstream gradient — keep a finite factor: 2^(8 - ex) must not pass 2^127)
                const float sc = ldexpf(1.0f, 8 - ex);                               // row max -> [128, 256)
                if (wave == 0) s_rinv[lane] = ldexpf(1.0f, ex - 8);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float os[4] = {oall[u][0] * sc, oall[u][1] * sc, oall[u][2] * sc, oall[u][3] * sc};
                    store_quad_xh<true, X3_LDP>(X, lane, 64 * wave + 4 * u, os);
                }
            }
        }
        const int next_tile = tile + (int)gridDim.x;
        const int qn = next_tile * 64 + lane;
        int n_p = -1, n_srow = 0, n_off = 0, n_idx = -1;
        float n_w = 0.f;
        if (qn < NP) {
            n_p = pair_point[qn];
            n_w = wn[qn];
        }
        T_MARK(16)
        lds_barrier();
        T_MARK(17)
        f32x16 acc[2][2];
        f32x16 accc[2][H2 ? 2 : 1];        // H2: the cross terms' accumulators
        const int j = lane & 31;
        uint32_t mw[2] = {mk[512 + (2 * wave) * 64 + lane], mk[512 + (2 * wave + 1) * 64 + lane]};      // layer-2 sign words, before the GEMM
        zero_acc(acc);
        WFrag3 nf = cx_gemm<CX_TH, false, H2>(X, w_bw3, lane, acc, fr3, w_bw2, accc);
        T_MARK(18)
        lds_barrier();
        T_MARK(19)
        cx_bwd_epilogue<H2>(X, acc, wave, lane, mw, G2 + tbase, s_rinv);
        T_MARK(20)
        lds_barrier();
        T_MARK(21)
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        T_MARK(22)
        mw[0] = mk[(2 * wave) * 64 + lane];
        mw[1] = mk[(2 * wave + 1) * 64 + lane];
        zero_acc(acc);
        cx_gemm<CX_TH, false, H2>(X, w_bw2, lane, acc, nf, nullptr, accc);
        T_MARK(18)
        gx3 w_bwl = frag + CX_BWL + (wave >> 1) * (CX_TH * 3 * 64) + lane;
        const WFrag1 frl = load_wfrag1(w_bwl);
        // third link of the next tile's lookup chain (pair -> point -> slot -> NEIGHBOUR), ahead of the epilogue that covers its round trip
        if (n_p >= 0) n_idx = nbr[(size_t)n_srow * k + (qn - n_off)];
        lds_barrier();
        T_MARK(19)
        cx_bwd_epilogue<H2>(X, acc, wave, lane, mw, G1 + tbase, s_rinv);
        T_MARK(20)
        lds_barrier();
        T_MARK(21)
        if (next_tile < ntiles) cur = cx_fetch_grow(g_agg3, masks + (size_t)next_tile * 3 * 512, n_p, n_idx, n_w, lane, wave);
        T_MARK(22)
        // ---- d/d latent = G1 W0[:, 39:103]: wave = (latent half m, row half n), one 32x32 tile each; scatter-add -------------------
        {
            const int m = wave >> 1, n = wave & 1, kg = lane >> 5;
            f32x16 aj = gemm_x3_tile<CX_TH, X3_LDP, H2>(X, n, w_bwl, lane, frl);
            if (H2) aj = aj * s_rinv[32 * n + j];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(&L[(32 * n + j) * CX_LDL + 32 * m + 8 * g + 4 * kg]) = f32x4{aj[4 * g], aj[4 * g + 1], aj[4 * g + 2], aj[4 * g + 3]};
        }
        T_MARK(25)
        lds_barrier();
        // Rows of a tile that hit the same neural point (samples along a ray share most of their neighbours) are summed in LDS
        // first: same-address atomics serialise in L2, and with one workgroup per CU nothing else runs meanwhile.
        // thread = (row, quarter of the 64 latent columns): the first row of each group of equal indices takes the group's sum (in
        // place: a row is read by its group's first row only), all groups in parallel.
        {
            const int my = s_idx[row0];
            uint32_t part = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int4 v = *reinterpret_cast<const int4*>(&s_idx[16 * q40 + 4 * u]);
                part |= (v.x == my ? 1u : 0u) << (4 * u) | (v.y == my ? 2u : 0u) << (4 * u) | (v.z == my ? 4u : 0u) << (4 * u) |
                        (v.w == my ? 8u : 0u) << (4 * u);
            }
            unsigned long long mask = (unsigned long long)part << (16 * q40);
            mask |= __shfl_xor(mask, 1);
            mask |= __shfl_xor(mask, 2);
            const bool leader = my >= 0 && (mask & ((1ull << row0) - 1ull)) == 0ull;
            if (q40 == 0) s_lead[row0] = leader ? my : -1;
            mask &= ~(1ull << row0);
            if (leader && mask) {
                float* own = &L[row0 * CX_LDL + 16 * q40];
                f32x4 sum[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) sum[u] = *reinterpret_cast<const f32x4*>(own + 4 * u);
                while (mask) {
                    const int r2 = __builtin_ctzll(mask);
                    mask &= mask - 1ull;
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum[u] += *reinterpret_cast<const f32x4*>(&L[r2 * CX_LDL + 16 * q40 + 4 * u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(own + 4 * u) = sum[u];
            }
        }
        T_MARK(26)
        lds_barrier();
        // wave w adds rows 16 w .. 16 w + 15 that lead a group: one row per instruction, lane = latent column (256 contiguous bytes per
        // atomic instruction; a lane-per-row arrangement touches 32 to 64 cache lines per instruction and runs at a fraction of the
        // rate).  The 16 row reads are independent of each other and of the atomics.
        {
            int lead[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int4 v = *reinterpret_cast<const int4*>(&s_lead[16 * wave + 4 * u]);
                lead[4 * u] = v.x; lead[4 * u + 1] = v.y; lead[4 * u + 2] = v.z; lead[4 * u + 3] = v.w;
            }
            float val[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) val[rr] = L[(16 * wave + rr) * CX_LDL + lane];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int idx = __builtin_amdgcn_readfirstlane(lead[rr]);
                if (idx >= 0) {
                    if (g_fixed) fixed_add(g_fixed, (size_t)idx * SPF_COL_DIM + lane, val[rr]);      // order-independent (common.h)
                    else atomicAdd(g_feat_col + (size_t)idx * SPF_COL_DIM + lane, val[rr]);
                }
            }
        }
        T_MARK(23)
        lds_barrier();
        T_MARK(24)
    }
    T_FLUSH
}


// both images in one launch (the weights change every optimisation step: the packing is on the step's critical path)
__global__ void color_pack_kernel(CPackArgs a, float* __restrict__ out, float* __restrict__ zero_buf, long long zero_floats) {
    color_pack_all(a, out, zero_buf, zero_floats, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

}  // namespace

SPF_DEFINE_TIMING_ENTRY(spf_debug_timing_color)

extern "C" {

int64_t spf_color_packed_floats(void) { return C_PACKED_TOTAL; }

int spf_color_pack(const float* w0, const float* b0, const float* w2, const float* b2, const float* w4, const float* b4,
                   float* packed, float* zero_buf, int64_t zero_floats, void* stream) {
    if (!w0 || !b0 || !w2 || !b2 || !w4 || !b4 || !packed) return spf::fail(SPF_EINVAL, "spf_color_pack: null pointer");
    if (zero_floats < 0 || (zero_buf && ((uintptr_t)zero_buf & 15))) return spf::fail(SPF_EINVAL, "spf_color_pack: zero_buf must be 16-byte aligned, zero_floats >= 0");
    CPackArgs a{w0, b0, w2, b2, w4, b4};
    color_pack_kernel<<<spf::div_up(C_PACK_THREADS, 256), 256, 0, (hipStream_t)stream>>>(a, packed, zero_buf, (long long)zero_floats);
    SPF_LAUNCH_CHECK("color_pack_kernel");
    return SPF_OK;
}

int spf_color_forward(const float* x, const int32_t* nbr, const float* wn, const int32_t* point_slot, const int32_t* pair_off,
                      const int32_t* pair_point, const int32_t* n_pairs, int32_t max_pairs, int32_t k, const float* pts,
                      const float* feat_color, const float* packed, float* agg3, float* act0, float* act1, float* act2, uint32_t* masks,
                      int64_t* agg3_fixed, int32_t arith, void* stream) {
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32 && arith != SPF_ARITH_H2)
        return spf::fail(SPF_EINVAL, "spf_color_forward: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (max_pairs < 0 || k < 1 || k > SPF_KMAX) return spf::fail(SPF_EINVAL, "spf_color_forward: bad sizes");
    if (max_pairs == 0) return SPF_OK;
    if (!x || !nbr || !wn || !pair_off || !pair_point || !pts || !feat_color || !packed || !agg3)
        return spf::fail(SPF_EINVAL, "spf_color_forward: null pointer");
    const bool store = act0 != nullptr;
    if (store && (!act1 || !act2 || !masks)) return spf::fail(SPF_EINVAL, "spf_color_forward: training buffers must be given together");
    if (agg3_fixed && arith == SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_color_forward: the fixed-point accumulator needs SPF_ARITH_SPLIT or SPF_ARITH_H2");
    long long* afx = reinterpret_cast<long long*>(agg3_fixed);
    const int tiles = spf::div_up(max_pairs, 64);
    const int blocks = tiles < 512 ? tiles : 512;
    if (arith == SPF_ARITH_H2) {
        const int b1 = tiles < 256 ? tiles : 256;
        if (store)
            color_forward_x3_kernel<true, true><<<b1, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts,
                                                                                     feat_color, packed, agg3, act0, act1, act2, masks, afx);
        else
            color_forward_x3_kernel<false, true><<<b1, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k,
                                                                                      pts, feat_color, packed, agg3, nullptr, nullptr, nullptr, nullptr, afx);
        SPF_LAUNCH_CHECK("color_forward_x3_kernel<H2>");
        return SPF_OK;
    }
    if (arith == SPF_ARITH_SPLIT) {
        const int b1 = tiles < 256 ? tiles : 256;   // one workgroup per CU (bf16 planes: 101 KB of LDS)
        if (store)
            color_forward_x3_kernel<true><<<b1, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts,
                                                                               feat_color, packed, agg3, act0, act1, act2, masks, afx);
        else
            color_forward_x3_kernel<false><<<b1, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k,
                                                                                pts, feat_color, packed, agg3, nullptr, nullptr, nullptr, nullptr, afx);
        SPF_LAUNCH_CHECK("color_forward_x3_kernel");
        return SPF_OK;
    }
    if (store)
        color_forward_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts,
                                                                            feat_color, packed, agg3, act0, act1, act2, masks);
    else
        color_forward_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(x, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k,
                                                                             pts, feat_color, packed, agg3, nullptr, nullptr, nullptr, nullptr);
    SPF_LAUNCH_CHECK("color_forward_kernel");
    return SPF_OK;
}

int spf_color_backward(const float* g_agg3, const int32_t* nbr, const float* wn, const int32_t* point_slot, const int32_t* pair_off,
                       const int32_t* pair_point, const int32_t* n_pairs, int32_t max_pairs, int32_t k, const float* packed,
                       const uint32_t* masks, float* G1, float* G2, float* G3, float* g_b0, float* g_b2, float* g_b4, float* g_feat_color,
                       int64_t* g_feat_color_fixed, int32_t arith, void* stream) {
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32 && arith != SPF_ARITH_H2)
        return spf::fail(SPF_EINVAL, "spf_color_backward: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (max_pairs < 0 || k < 1 || k > SPF_KMAX) return spf::fail(SPF_EINVAL, "spf_color_backward: bad sizes");
    if (max_pairs == 0) return SPF_OK;
    if (!g_agg3 || !nbr || !wn || !pair_off || !pair_point || !packed || !masks || !G1 || !G2 || !G3 || !g_b0 || !g_b2 || !g_b4 ||
        (!g_feat_color && !g_feat_color_fixed))
        return spf::fail(SPF_EINVAL, "spf_color_backward: null pointer");
    if (g_feat_color_fixed && arith == SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_color_backward: the fixed-point latent accumulator needs SPF_ARITH_SPLIT or SPF_ARITH_H2");
    const int tiles = spf::div_up(max_pairs, 64);
    const int blocks = tiles < 512 ? tiles : 512;
    if (arith == SPF_ARITH_SPLIT || arith == SPF_ARITH_H2) {    // bias gradients come from spf_wgrad (dbias) in these modes: g_b0 / g_b2 / g_b4 are not touched
        const int b1 = tiles < 256 ? tiles : 256;
        if (arith == SPF_ARITH_H2)
            color_backward_x3_kernel<true><<<b1, 256, 0, (hipStream_t)stream>>>(g_agg3, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, packed,
                                                                                masks, G1, G2, G3, g_feat_color, reinterpret_cast<long long*>(g_feat_color_fixed));
        else
            color_backward_x3_kernel<false><<<b1, 256, 0, (hipStream_t)stream>>>(g_agg3, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, packed,
                                                                                 masks, G1, G2, G3, g_feat_color, reinterpret_cast<long long*>(g_feat_color_fixed));
        SPF_LAUNCH_CHECK("color_backward_x3_kernel");
        return SPF_OK;
    }
    color_backward_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g_agg3, nbr, wn, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, packed,
                                                                   masks, G1, G2, G3, g_b0, g_b2, g_b4, g_feat_color);
    SPF_LAUNCH_CHECK("color_backward_kernel");
    return SPF_OK;
}

}  // extern "C"
