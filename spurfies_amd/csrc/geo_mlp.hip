// Fused geometry path (K4 of DESIGN.md): gather 32-d geometry latents + relative positions of a
// point's <= 8 neighbours, RBF weights, F_geometry + T on the fp32 matrix cores, the input-Jacobian
// sweep, and the RBF-weighted mean per point — one kernel, activations never leave the CU.
//
// Replaces, for every caller, the reference's
//   get_keypoint_data  spurfies/model/utils.py:140-170       (table concat + index_select + masked_select)
//   compute_weights    spurfies/model/pointneus_disent.py:241-247
//   get_sdf            spurfies/model/pointneus_disent.py:300-313 (5+1 cuBLAS GEMMs, index_add_)
//   get_gradients      spurfies/model/pointneus_disent.py:315-323 (autograd double-backward graph)
// and the inlined copies in sdf_importance :386-418, pseudo_sdf :460-493, get_sdf_eval :284-296.
//
// Shape of the work: one tile = 8 points x 8 neighbour slots = 64 rows.  Four waves per workgroup;
// wave w owns output columns [64w, 64w+64) of every 256-wide layer as 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).  The A operand (activations,
// [64][256] fp32, row stride 260 floats so ds_read_b128 is conflict-free) lives in LDS; the B operand
// (weights) streams from L2 in a fragment order packed once by spf_geo_pack, 1 KiB per wave-load.
//
// Math (LeakyReLU slope 0.01 = nn.LeakyReLU default):
//   h1 = W0 [g | x_pi] + b0, a1 = lrelu(h1), ..., a4 = lrelu(W6 a3 + b6)
//   sdf_j = T (W8 a4 + b8) + bT = v . a4 + c,  v = T W8, c = T b8 + bT   (no activation between
//           F_geometry's last Linear and T, pointneus_disent.py:95-98, so they fold exactly in R)
//   d sdf_j / d in = (((v * D4) W6 * D3) W4 * D2) W2 * D1) W0,  D_l = lrelu'(h_l) in {1, 0.01}
//   sdf(p) = sum_j w_j sdf_j / sum_j w_j,  w_j = exp(-(rbf * max(|x_pi|, 1e-12))^2)  (detached)
//   d sdf / d x(p) = sum_j (w_j / norm) d sdf_j / d x_pi
#include "grid_dev.h"
#include "mlp_tile.h"
#include "mlp_tile_x3s.h"

namespace {

using namespace spf;

constexpr int K_IN = 35;     // 32 latent + 3 x_pi
constexpr int T_IN = 5;      // ceil(35/8) k-steps of 8 for the first layer

// packed image layout (floats)
constexpr int SZ_FW1 = 4 * T_IN * 2 * 64 * 4;
constexpr int SZ_HH = 4 * T_HID * 2 * 64 * 4;
constexpr int SZ_JW = 2 * T_HID * 64 * 4;
constexpr int OFF_FW1 = 0;
constexpr int OFF_FW2 = OFF_FW1 + SZ_FW1;
constexpr int OFF_FW3 = OFF_FW2 + SZ_HH;
constexpr int OFF_FW4 = OFF_FW3 + SZ_HH;
constexpr int OFF_BW4 = OFF_FW4 + SZ_HH;  // d/d a3 = g_h4 * W6
constexpr int OFF_BW3 = OFF_BW4 + SZ_HH;  // W4
constexpr int OFF_BW2 = OFF_BW3 + SZ_HH;  // W2
constexpr int OFF_JW1 = OFF_BW2 + SZ_HH;  // W0 (256 -> 35, padded to 64)
constexpr int OFF_B1 = OFF_JW1 + SZ_JW;
constexpr int OFF_B2 = OFF_B1 + 256;
constexpr int OFF_B3 = OFF_B2 + 256;
constexpr int OFF_B4 = OFF_B3 + 256;
constexpr int OFF_V5 = OFF_B4 + 256;
constexpr int OFF_C = OFF_V5 + 256;
constexpr int PACKED_FLOATS = OFF_C + 4;

// LDS: the activation tile only
constexpr int L_X = 0;
constexpr int L_TOTAL = 64 * LDA;

// per-pair scratch written by the MLP kernel and consumed by the point reduction: [w, sdf_j, dsdf_j/dx (3)]
constexpr int PT_STRIDE = spf::GEO_PT_STRIDE;

// forward epilogue: + bias, record sign bits, LeakyReLU, write this wave's 64x64 block back to X
__device__ __forceinline__ void fwd_epilogue(float* X, const f32x16 (&acc)[2][2], const float (&bv)[2], int wave,
                                             int lane, uint32_t (&mask)[2]) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
    mask[0] = mask[1] = 0u;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] + bv[n];
                const bool pos = v > 0.f;
                mask[m] |= (pos ? 1u : 0u) << (n * 16 + r);
                v = pos ? v : v * 0.01f;
                X[(m * 32 + row_of(r, h)) * LDA + c0 + 32 * n] = v;
            }
}

// backward epilogue: g_h = g_a * lrelu'(h)
__device__ __forceinline__ void bwd_epilogue(float* X, const f32x16 (&acc)[2][2], int wave, int lane, const uint32_t (&mask)[2]) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool pos = (mask[m] >> (n * 16 + r)) & 1u;
                const float v = acc[m][n][r];
                X[(m * 32 + row_of(r, h)) * LDA + c0 + 32 * n] = pos ? v : v * 0.01f;
            }
}

// One tile = 64 consecutive VALID pairs (rows), whatever points they belong to: no padding rows except in
// the last tile.  Per pair the kernel leaves [w_j, sdf_j, d sdf_j/d x] in pair_tmp and (WITH_JAC) the
// latent Jacobian row in jac; geo_point_reduce_kernel then forms the per-point weighted means.
template <bool WITH_JAC>
__global__ void __launch_bounds__(256, 2)
geo_pairs_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot,
                 const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point, const int32_t* __restrict__ n_pairs_dev,
                 int max_pairs, int k, const float* __restrict__ pts, const float* __restrict__ feat_geo, const float* packed,
                 float rbf, float* __restrict__ pair_tmp, float* __restrict__ jac) {
    __shared__ __attribute__((aligned(16))) float smem[L_TOTAL];
    float* X = smem + L_X;
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
        T_MARK(31)
        gf4p pk4 = reinterpret_cast<gf4p>(packed);
        const BFrag fr1 = load_bfrag(pk4 + (OFF_FW1 / 4) + wave * (T_IN * 128), lane);     // in flight during the gather
        const int qn = (tile + (int)gridDim.x) * 64 + (tid >> 2);                          // this thread's row in the block's next tile
        const bool has_next = tile + (int)gridDim.x < ntiles && qn < NP;
        // ---- gather: thread = (row, quarter of the 32-d latent) ---------------------------------
        {
            const int row = tid >> 2, q4 = tid & 3;
            const int q = tile * 64 + row;
            const int idx = n_idx, srow = n_srow;
            f32x4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = f0;
            if (idx >= 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(feat_geo + (size_t)idx * SPF_GEO_DIM + q4 * 8);
                f0 = src[0];
                f1 = src[1];
            }
            *reinterpret_cast<f32x4*>(X + row * LDA + q4 * 8) = f0;
            *reinterpret_cast<f32x4*>(X + row * LDA + q4 * 8 + 4) = f1;
            if (q4 == 0) {
                float dx = 0.f, dy = 0.f, dz = 0.f;
                if (idx >= 0) {
                    dx = x[(size_t)srow * 3] - pts[(size_t)idx * 3];
                    dy = x[(size_t)srow * 3 + 1] - pts[(size_t)idx * 3 + 1];
                    dz = x[(size_t)srow * 3 + 2] - pts[(size_t)idx * 3 + 2];
                    const float dist = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
                    const float sc = dist * rbf;
                    pair_tmp[(size_t)q * PT_STRIDE] = expf(-(sc * sc));
                }
                *reinterpret_cast<f32x4*>(X + row * LDA + 32) = f32x4{dx, dy, dz, 0.f};
                *reinterpret_cast<f32x4*>(X + row * LDA + 36) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        T_MARK(0)
        __syncthreads();
        T_MARK(1)

        f32x16 acc[2][2];
        uint32_t m1[2], m2[2], m3[2], m4[2];
        // ---- forward: 35 -> 256 -> 256 -> 256 -> 256 -------------------------------------------
        // each layer's bias values are requested before its GEMM and the next layer's first weight fragment inside it, so
        // neither L2 round trip is exposed between the barriers
        const int cb = wave * 64 + (lane & 31);
        gf4p wfw1 = pk4 + (OFF_FW1 / 4) + wave * (T_IN * 128);
        gf4p wfw2 = pk4 + (OFF_FW2 / 4) + wave * (T_HID * 128);
        gf4p wfw3 = pk4 + (OFF_FW3 / 4) + wave * (T_HID * 128);
        gf4p wfw4 = pk4 + (OFF_FW4 / 4) + wave * (T_HID * 128);
        gf4p wbw4 = pk4 + (OFF_BW4 / 4) + wave * (T_HID * 128);
        gf4p wbw3 = pk4 + (OFF_BW3 / 4) + wave * (T_HID * 128);
        gf4p wbw2 = pk4 + (OFF_BW2 / 4) + wave * (T_HID * 128);
        float bv[2] = {packed[OFF_B1 + cb], packed[OFF_B1 + cb + 32]};
        n_p = has_next ? pair_point[qn] : -1;
        zero_acc(acc);
        BFrag nf = gemm_rows64<T_IN>(X, wfw1, lane, acc, fr1, wfw2);
        T_MARK(2)
        __syncthreads();
        T_MARK(3)
        fwd_epilogue(X, acc, bv, wave, lane, m1);
        T_MARK(4)
        __syncthreads();
        T_MARK(5)
        bv[0] = packed[OFF_B2 + cb]; bv[1] = packed[OFF_B2 + cb + 32];
        int n_off = 0;
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        zero_acc(acc);
        nf = gemm_rows64<T_HID>(X, wfw2, lane, acc, nf, wfw3);
        T_MARK(2)
        __syncthreads();
        T_MARK(3)
        fwd_epilogue(X, acc, bv, wave, lane, m2);
        T_MARK(4)
        __syncthreads();
        T_MARK(5)
        bv[0] = packed[OFF_B3 + cb]; bv[1] = packed[OFF_B3 + cb + 32];
        n_idx = n_p >= 0 ? nbr[(size_t)n_srow * k + (qn - n_off)] : -1;
        zero_acc(acc);
        nf = gemm_rows64<T_HID>(X, wfw3, lane, acc, nf, wfw4);
        T_MARK(2)
        __syncthreads();
        T_MARK(3)
        fwd_epilogue(X, acc, bv, wave, lane, m3);
        T_MARK(4)
        __syncthreads();
        T_MARK(5)
        bv[0] = packed[OFF_B4 + cb]; bv[1] = packed[OFF_B4 + cb + 32];
        const float vv[2] = {packed[OFF_V5 + cb], packed[OFF_V5 + cb + 32]};       // folded last layer, used by the sweep
        zero_acc(acc);
        nf = gemm_rows64<T_HID>(X, wfw4, lane, acc, nf, WITH_JAC ? wbw4 : nullptr);
        T_MARK(2)
        __syncthreads();
        T_MARK(3)
        fwd_epilogue(X, acc, bv, wave, lane, m4);
        T_MARK(4)
        __syncthreads();
        T_MARK(5)

        // ---- sdf_j = v . a4 + c : 4 threads per row, interleaved float4 chunks ------------------
        {
            const int row = tid >> 2, q4 = tid & 3;
            gf4p v4 = pk4 + OFF_V5 / 4;
            float s = 0.f;
#pragma unroll
            for (int mth = 0; mth < 16; ++mth) {
                const int c4 = q4 + 4 * mth;
                const f32x4 a = *reinterpret_cast<const f32x4*>(X + row * LDA + 4 * c4);
                const f32x4 v = v4[c4];
                s += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            const int q = tile * 64 + row;
            if (q4 == 0 && q < NP) pair_tmp[(size_t)q * PT_STRIDE + 1] = s + packed[OFF_C];
        }

        T_MARK(6)
        if (WITH_JAC) {
            __syncthreads();
            T_MARK(7)
            // ---- Jacobian sweep: g_h4 = v * D4 ; g_a3 = g_h4 W6 ; ... ; J = g_h1 W0 --------------
            {
                const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const bool pos = (m4[m] >> (n * 16 + r)) & 1u;
                            X[(m * 32 + row_of(r, h)) * LDA + c0 + 32 * n] = pos ? vv[n] : vv[n] * 0.01f;
                        }
            }
            T_MARK(8)
            __syncthreads();
            T_MARK(9)
            zero_acc(acc);
            nf = gemm_rows64<T_HID>(X, wbw4, lane, acc, nf, wbw3);
            T_MARK(10)
            __syncthreads();
            T_MARK(11)
            bwd_epilogue(X, acc, wave, lane, m3);
            T_MARK(12)
            __syncthreads();
            T_MARK(13)
            zero_acc(acc);
            nf = gemm_rows64<T_HID>(X, wbw3, lane, acc, nf, wbw2);
            T_MARK(10)
            __syncthreads();
            T_MARK(11)
            bwd_epilogue(X, acc, wave, lane, m2);
            T_MARK(12)
            __syncthreads();
            T_MARK(13)
            zero_acc(acc);
            gemm_rows64<T_HID>(X, wbw2, lane, acc, nf, nullptr);
            T_MARK(10)
            __syncthreads();
            T_MARK(11)
            bwd_epilogue(X, acc, wave, lane, m1);
            T_MARK(12)
            __syncthreads();
            T_MARK(13)
            // last step 256 -> 35 (padded 64): wave = (row half mt, column half nt), one 32x32 tile each
            {
                const int mt = wave >> 1, nt = wave & 1, i = lane & 31, h = lane >> 5;
                f32x16 aj;
#pragma unroll
                for (int r = 0; r < 16; ++r) aj[r] = 0.f;
                const float* ap = X + (mt * 32 + i) * LDA + 4 * h;
                gf4p bp = pk4 + (OFF_JW1 / 4) + nt * (T_HID * 64) + lane;
#pragma unroll 4
                for (int t = 0; t < T_HID; ++t) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * t);
                    const f32x4 b = bp[t * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) aj = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], aj, 0, 0, 0);
                }
                const int qb = tile * 64 + mt * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qb + (r & 3) + 8 * (r >> 2);
                    if (q < NP) {
                        if (nt == 0) jac[(size_t)q * SPF_GEO_DIM + i] = aj[r];
                        else if (i < 3) pair_tmp[(size_t)q * PT_STRIDE + 2 + i] = aj[r];
                    }
                }
            }
        }
        T_MARK(14)
        __syncthreads();  // smem is reused by the next tile
        T_MARK(15)
    }
    T_FLUSH
}

// per point: norm = sum_j w_j ; sdf = sum_j w_j sdf_j / norm ; wn_j = w_j / norm ; grad = sum_j wn_j d sdf_j/dx
__global__ void geo_point_reduce_kernel(const float* __restrict__ pair_tmp, const int32_t* __restrict__ pair_off,
                                        const int32_t* __restrict__ point_slot, const int32_t* __restrict__ n_points_dev, int max_points,
                                        float* __restrict__ sdf, float* __restrict__ grad, float* __restrict__ wn) {
    const int P = n_points_dev ? min(*n_points_dev, max_points) : max_points;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int q0 = pair_off[p], q1 = pair_off[p + 1];
    const int srow = point_slot ? point_slot[p] : p;
    float nrm = 0.f, a = 0.f;
    for (int q = q0; q < q1; ++q) {
        const float w = pair_tmp[(size_t)q * PT_STRIDE];
        nrm += w;
        a += w * pair_tmp[(size_t)q * PT_STRIDE + 1];
    }
    sdf[srow] = a / nrm;
    const float inv = 1.0f / nrm;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int q = q0; q < q1; ++q) {
        const float w = pair_tmp[(size_t)q * PT_STRIDE];
        if (wn) wn[q] = w / nrm;
        if (grad) {
            const float wi = w * inv;
            g0 += wi * pair_tmp[(size_t)q * PT_STRIDE + 2];
            g1 += wi * pair_tmp[(size_t)q * PT_STRIDE + 3];
            g2 += wi * pair_tmp[(size_t)q * PT_STRIDE + 4];
        }
    }
    if (grad) {
        grad[(size_t)srow * 3] = g0;
        grad[(size_t)srow * 3 + 1] = g1;
        grad[(size_t)srow * 3 + 2] = g2;
    }
}

// g_feat[nbr(q)] += g_sdf[row(q)] * wn[q] * jac[q, :].  One workgroup per tile of 64 consecutive pairs (pairs are grouped by point,
// points follow each other along a ray: neighbouring samples share most of their neighbours, so a tile holds each neural point 2-4
// times).  Rows of a tile that hit the SAME neural point are summed in LDS first (fixed order: ascending row) and each group is added
// with ONE 128-byte atomic row — same-address atomics serialise in L2, and atomics are priced per cache line touched.
__global__ void __launch_bounds__(256)
geo_backward_latents_kernel(const float* __restrict__ g_sdf, const float* __restrict__ wn, const float* __restrict__ jac,
                            const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off,
                            const int32_t* __restrict__ pair_point, const int32_t* __restrict__ n_pairs_dev, int max_pairs, int k,
                            float* __restrict__ g_feat, long long* __restrict__ g_fixed, const float* __restrict__ grad_x,
                            float* __restrict__ g_x, int n_rows) {
    if (g_x) {      // d L / d x[row] = g_sdf[row] * d sdf / d x[row] (the RBF weights are detached): rides along, one element per thread
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 3 * n_rows; e += gridDim.x * blockDim.x) g_x[e] = g_sdf[e / 3] * grad_x[e];
    }
    constexpr int LDL = SPF_GEO_DIM + 4;                                  // row stride of L in floats (16-byte aligned, off the bank period)
    __shared__ __attribute__((aligned(16))) float L[64 * LDL];
    __shared__ __attribute__((aligned(16))) int s_idx[64];
    __shared__ __attribute__((aligned(16))) int s_lead[64];
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = tid >> 2, q40 = tid & 3;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- thread = (pair row, quarter of the 32 latent columns): coef * jac row -> LDS
        {
            const int q = tile * 64 + row0;
            int idx = -1;
            float coef = 0.f;
            if (q < NP) {
                const int p = pair_point[q];
                const int srow = point_slot ? point_slot[p] : p;
                idx = nbr[(size_t)srow * k + (q - pair_off[p])];
                coef = g_sdf[srow] * wn[q];
            }
            if (coef == 0.f) idx = -1;                                    // nothing to add (also rows past the pair count)
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (idx >= 0) {
                const f32x4* src = reinterpret_cast<const f32x4*>(jac + (size_t)q * SPF_GEO_DIM + 8 * q40);
                a = src[0] * coef;
                b = src[1] * coef;
            }
            *reinterpret_cast<f32x4*>(&L[row0 * LDL + 8 * q40]) = a;
            *reinterpret_cast<f32x4*>(&L[row0 * LDL + 8 * q40 + 4]) = b;
            if (q40 == 0) s_idx[row0] = idx;
        }
        __syncthreads();
        // ---- the first row of each group of equal indices takes the group's sum, all groups in parallel
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
            if (leader && mask) {                                         // in place: a row is read by its group's first row only
                float* own = &L[row0 * LDL + 8 * q40];
                f32x4 sa = *reinterpret_cast<const f32x4*>(own), sb = *reinterpret_cast<const f32x4*>(own + 4);
                while (mask) {
                    const int r2 = __builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    sa += *reinterpret_cast<const f32x4*>(&L[r2 * LDL + 8 * q40]);
                    sb += *reinterpret_cast<const f32x4*>(&L[r2 * LDL + 8 * q40 + 4]);
                }
                *reinterpret_cast<f32x4*>(own) = sa;
                *reinterpret_cast<f32x4*>(own + 4) = sb;
            }
        }
        __syncthreads();
        // ---- wave w adds the leading rows among 16 w .. 16 w + 15, two rows per instruction (lane = (row parity, column))
        {
            const int c = lane & 31, half = lane >> 5;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = 16 * wave + 2 * rr + half;
                const int idx = s_lead[row];
                if (idx >= 0) {
                    const float v = L[row * LDL + c];
                    if (g_fixed) fixed_add(g_fixed, (size_t)idx * SPF_GEO_DIM + c, v);       // order-independent (common.h)
                    else atomicAdd(&g_feat[(size_t)idx * SPF_GEO_DIM + c], v);
                }
            }
        }
        __syncthreads();                                                  // L / s_idx are rewritten by the next tile
    }
}

struct PackArgs {
    const float *w0, *b0, *w2, *b2, *w4, *b4, *w6, *b6, *w8, *b8, *wT, *bT;
};

__global__ void geo_pack_kernel(PackArgs a, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= PACKED_FLOATS) return;
    float val = 0.f;
    if (e < OFF_JW1) {
        // [wave][t][nt][lane][j] fragments; forward layers read W[n][k], backward layers W[k][n]
        int region, local;
        if (e < OFF_FW2) { region = 0; local = e; }
        else { region = 1 + (e - OFF_FW2) / SZ_HH; local = (e - OFF_FW2) % SZ_HH; }
        const int T = region == 0 ? T_IN : T_HID;
        const int j = local & 3, ln = (local >> 2) & 63, nt = (local >> 8) & 1;
        const int t = (local >> 9) % T, w = (local >> 9) / T;
        const int n = 64 * w + 32 * nt + (ln & 31), kk = 8 * t + 4 * (ln >> 5) + j;
        switch (region) {
            case 0: val = kk < K_IN ? a.w0[n * K_IN + kk] : 0.f; break;
            case 1: val = a.w2[n * 256 + kk]; break;
            case 2: val = a.w4[n * 256 + kk]; break;
            case 3: val = a.w6[n * 256 + kk]; break;
            case 4: val = a.w6[kk * 256 + n]; break;  // g_a3[i] = sum_o g_h4[o] W6[o][i]
            case 5: val = a.w4[kk * 256 + n]; break;
            default: val = a.w2[kk * 256 + n]; break;
        }
    } else if (e < OFF_B1) {
        const int local = e - OFF_JW1;
        const int j = local & 3, ln = (local >> 2) & 63, t = (local >> 8) & 31, nt = (local >> 13) & 1;
        const int n = 32 * nt + (ln & 31), kk = 8 * t + 4 * (ln >> 5) + j;
        val = n < K_IN ? a.w0[kk * K_IN + n] : 0.f;
    } else if (e < OFF_V5) {
        const int local = e - OFF_B1, l = local >> 8, i = local & 255;
        val = (l == 0 ? a.b0 : l == 1 ? a.b2 : l == 2 ? a.b4 : a.b6)[i];
    } else if (e < OFF_C) {
        const int i = e - OFF_V5;
        float s = 0.f;
        for (int o = 0; o < 256; ++o) s += a.wT[o] * a.w8[o * 256 + i];
        val = s;
    } else if (e == OFF_C) {
        float s = 0.f;
        for (int o = 0; o < 256; ++o) s += a.wT[o] * a.b8[o];
        val = s + a.bT[0];
    }
    out[e] = val;
}


// ==============================================================================================================================
// Same path with fp32-CLASS products (six exact bf16 piece products, <= 2 ulp per fp32 product) from three bf16 pieces per operand (arith = SPF_ARITH_SPLIT / SPF_ARITH_SPLIT_W; rounds 1 - 5's default.  Round 6's default is H2 — three exact fp16 piece products from two fp16 pieces per operand — on the 32x32x16 engine further down: SPF_ARITH_H2.  arith = SPF_ARITH_F32 selects the kernel above).
//   x = p1 + p2 + p3, p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2): both differences are exact in fp32 and 3 x 8
//   mantissa bits cover fp32's 24; a product of two fp32 numbers is sum_{i,j} a_i b_j, every piece product is exact in the
//   MFMA's fp32 accumulation and the three with i + j >= 5 are below 2^-24 of the product, so the six with i + j <= 4 give the
//   fp32 product to fp32 rounding.  Six bf16 MFMAs (K = 16 in 32 cycles on a 32 x 32 tile) replace eight v_mfma_f32_32x32x2_f32
//   (64 cycles, K = 2): 2.7x the matrix rate; results differ from the fp32-MFMA kernel above only by summation order.
// Layout: transposed product D[feature][row] += W[feature][k] X[k][row] — the weights are the MFMA A operand (piece fragments streamed
// from L2), the activations the B operand, held in LDS as three bf16 planes [piece][row][k] (one 16-byte read per fragment); a lane
// owns 4 CONSECUTIVE features of a row per accumulator, so an epilogue (+ bias, LeakyReLU, sign bits, split into pieces) rewrites the
// planes with 8-byte stores.  One 4-wave workgroup per CU (the planes take 101 KB), wave w owns features [64w, 64w+64) x 64 rows.
// Round 2: the products run on v_mfma_f32_16x16x32_bf16 (mlp_tile_x3s.h: 4 x 4 tiles of 16 x 16, K = 32 per instruction, 16 cycles)
// instead of v_mfma_f32_32x32x16_bf16 (2 x 2 tiles, 32 cycles): same FLOP per cycle, same operand bytes, but the chip holds a higher
// clock under this shape (in the optimisation step: 1.99 vs 1.88 GHz on the same box, kernel -2.3 %; run back to back on its own:
// 2.05 vs 1.88 GHz, -5.2 %, at +3 % cycles).
// ==============================================================================================================================
// k32-steps of v_mfma_f32_16x16x32_bf16 (mlp_tile_x3s.h: 4 x 4 tiles of 16 x 16 per wave; the shape holds a higher clock under load)
constexpr int X3_T1 = 2;                          // first layer: K = 35 -> 64
constexpr int X3_TH = 8;                          // 256 / 32
constexpr int X3_SZ1 = 4 * X3_T1 * 4 * 3 * 64;    // bf16x8 entries: [wave][k32][a][piece][lane]
constexpr int X3_SZH = 4 * X3_TH * 4 * 3 * 64;
constexpr int X3_SZJ = 2 * X3_TH * 2 * 3 * 64;    // [m][k32][a][piece][lane]
constexpr int X3_FW1 = 0;
constexpr int X3_FW2 = X3_FW1 + X3_SZ1;
constexpr int X3_FW3 = X3_FW2 + X3_SZH;
constexpr int X3_FW4 = X3_FW3 + X3_SZH;
constexpr int X3_BW4 = X3_FW4 + X3_SZH;
constexpr int X3_BW3 = X3_BW4 + X3_SZH;
constexpr int X3_BW2 = X3_BW3 + X3_SZH;
constexpr int X3_JW1 = X3_BW2 + X3_SZH;
constexpr int X3_FRAGS = X3_JW1 + X3_SZJ;

// one thread per fragment slot (region, wave, k32, a, lane): 8 weights -> 3 x bf16x8.  lane = (i = lane & 15, g = lane >> 4):
// feature 16 a + i, k = 32 t + 8 g .. + 7
__global__ void geo_pack_x3_kernel(PackArgs a, bf16x8* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int N1 = X3_SZ1 / 3, NH = X3_SZH / 3, NJ = X3_SZJ / 3;
    if (s >= N1 + 6 * NH + NJ) return;
    int region, local;
    if (s < N1) { region = 0; local = s; }
    else if (s < N1 + 6 * NH) { region = 1 + (s - N1) / NH; local = (s - N1) % NH; }
    else { region = 7; local = s - N1 - 6 * NH; }
    const int ln = local & 63, i = ln & 15, g = ln >> 4;
    float w[8];
    size_t base;
    if (region < 7) {
        const int T = region == 0 ? X3_T1 : X3_TH;
        const int at = (local >> 6) & 3, t = (local >> 8) % T, wv = (local >> 8) / T;
        const int f = 64 * wv + 16 * at + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * t + 8 * g + e;
            float v;
            switch (region) {
                case 0: v = k < K_IN ? a.w0[f * K_IN + k] : 0.f; break;
                case 1: v = a.w2[f * 256 + k]; break;
                case 2: v = a.w4[f * 256 + k]; break;
                case 3: v = a.w6[f * 256 + k]; break;
                case 4: v = a.w6[k * 256 + f]; break;      // g_a3[f] = sum_o g_h4[o] W6[o][f]
                case 5: v = a.w4[k * 256 + f]; break;
                default: v = a.w2[k * 256 + f]; break;
            }
            w[e] = v;
        }
        const int rb = region == 0 ? X3_FW1 : X3_FW2 + (region - 1) * X3_SZH;
        base = (size_t)rb + (size_t)((wv * T + t) * 4 + at) * 3 * 64 + ln;
    } else {
        const int at = (local >> 6) & 1, t = (local >> 7) % X3_TH, m = (local >> 7) / X3_TH;
        const int f = 32 * m + 16 * at + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * t + 8 * g + e;
            w[e] = f < K_IN ? a.w0[k * K_IN + f] : 0.f;  // J[f] = sum_o g_h1[o] W0[o][f]
        }
        base = (size_t)X3_JW1 + (size_t)((m * X3_TH + t) * 2 + at) * 3 * 64 + ln;
    }
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 x, y, z;
        split3(w[e], x, y, z);
        p1[e] = x; p2[e] = y; p3[e] = z;
    }
    out[base] = p1;
    out[base + 64] = p2;
    out[base + 128] = p3;
}

// acc[a][b][e] = feature 64 w + 16 a + 4 g + e of row 16 b + i (i = lane & 15, g = lane >> 4).
// forward epilogue: a = lrelu(acc + b) -> planes; sign bits pushed into mask[b >> 1] in the order (a, b & 1, e) (32 per word).
// MODE 1 (last forward layer): sdf partial sums s[b] += v . a, and the planes receive the Jacobian seed v * lrelu'(h) instead of a
// (a itself is not needed any more).
template <int MODE, bool WITH_JAC>
__device__ __forceinline__ void fwd_epilogue_xs(__bf16* X, const f32x4 (&acc)[4][4], const BiasS& bias, const BiasS& v5, int wave, int lane,
                                                uint32_t (&mask)[2], float (&s)[4]) {
    const int i = lane & 15, g = lane >> 4;
    mask[0] = mask[1] = 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int f0 = 64 * wave + 16 * a + 4 * g;
        const f32x4 bv = bias.b[a];
        f32x4 vv = f32x4{0.f, 0.f, 0.f, 0.f}, vs = vv;
        if (MODE == 1) {
            vv = v5.b[a];
            vs = vv * 0.01f;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 h, hs;
            bias_scale4s(acc[a][b], bv, h, hs);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (MODE == 1) {
                    float seed;
                    const float act = lrelu_push_sel(h[e], hs[e], vv[e], vs[e], seed, mask[b >> 1]);
                    s[b] += vv[e] * act;
                    out[e] = seed;
                } else {
                    out[e] = lrelu_push(h[e], hs[e], mask[b >> 1]);
                }
            }
            if (MODE == 0 || WITH_JAC) store_quad_x3(X, 16 * b + i, f0, out);
        }
    }
}

// backward epilogue: g_h = g_a * lrelu'(h) -> planes (pops the words the forward epilogue filled, in the same order)
__device__ __forceinline__ void bwd_epilogue_xs(__bf16* X, const f32x4 (&acc)[4][4], int wave, int lane, const uint32_t (&mask_in)[2]) {
    const int i = lane & 15, g = lane >> 4;
    uint32_t mask[2] = {mask_in[0], mask_in[1]};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int f0 = 64 * wave + 16 * a + 4 * g;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 vs;
            scale4s(acc[a][b], vs);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = lrelu_pop(acc[a][b][e], vs[e], mask[b >> 1]);
            store_quad_x3(X, 16 * b + i, f0, out);
        }
    }
}

constexpr int X3_LDS_BF16 = 3 * X3_PLANE;

// Held-clock counters of the bf16-piece kernels, compiled in but OFF unless the call asks for them (arith | SPF_ARITH_CLOCK; the counters
// are one process-wide array per device, shared by every stream: a diagnostic for one measuring caller, documented as such in the header —
// a launch without the flag touches no global state).  With the flag (spf_geo_clock_read): thread 0 of every workgroup stamps the
// shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at kernel entry and exit and adds the two
// differences to a per-(MFMA shape, with / without Jacobian sweep) triple {cycles, ticks, workgroups}: sum(cycles) / sum(ticks) x 100 MHz
// is the shader clock the chip HELD while these kernels ran (DVFS give-back under MFMA-dense load, MI355X_MICROARCH.md) — measured in
// the run that reports it, not read from a file.  Cost: two scalar-memory reads and three atomics per workgroup per launch
// (256 workgroups x ~1.5 ms).
__device__ unsigned long long spf_geo_clock_buf[2][2][3];
#define CLK_DECL const unsigned long long clk_c0 = clk ? __builtin_amdgcn_s_memtime() : 0ull, clk_r0 = clk ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define CLK_FLUSH(ENGINE, JAC)                                                                            \
    if (clk && threadIdx.x == 0) {                                                                        \
        atomicAdd(&spf_geo_clock_buf[ENGINE][JAC][0], __builtin_amdgcn_s_memtime() - clk_c0);             \
        atomicAdd(&spf_geo_clock_buf[ENGINE][JAC][1], __builtin_amdgcn_s_memrealtime() - clk_r0);         \
        atomicAdd(&spf_geo_clock_buf[ENGINE][JAC][2], 1ull);                                              \
    }

// one pair row's gather operands: a quarter of the geometry latent, the offset x - p_i, the neighbour index
struct GxRow {
    f32x4 f0, f1;
    float d[3];
    int idx;
};
__device__ __forceinline__ GxRow gx_fetch_row(int idx, int srow, int q4, const float* __restrict__ x, const float* __restrict__ pts,
                                              const float* __restrict__ feat_geo) {
    GxRow r;
    r.idx = idx;
    r.f0 = r.f1 = f32x4{0.f, 0.f, 0.f, 0.f};
    r.d[0] = r.d[1] = r.d[2] = 0.f;
    if (idx >= 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(feat_geo + (size_t)idx * SPF_GEO_DIM + q4 * 8);
        r.f0 = src[0];
        r.f1 = src[1];
        r.d[0] = x[(size_t)srow * 3] - pts[(size_t)idx * 3];
        r.d[1] = x[(size_t)srow * 3 + 1] - pts[(size_t)idx * 3 + 1];
        r.d[2] = x[(size_t)srow * 3 + 2] - pts[(size_t)idx * 3 + 2];
    }
    return r;
}

template <bool WITH_JAC>
__global__ void __launch_bounds__(256, 1)
geo_pairs_x3_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot,
                    const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point, const int32_t* __restrict__ n_pairs_dev,
                    int max_pairs, int k, const float* __restrict__ pts, const float* __restrict__ feat_geo, const float* packed,
                    float rbf, float* __restrict__ pair_tmp, float* __restrict__ jac, int clk) {
    __shared__ __attribute__((aligned(16))) __bf16 X[X3_LDS_BF16];
    __shared__ float red[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    const int ntiles = (NP + 63) / 64;
    const float* packed0 = packed;
    T_DECL
    CLK_DECL
    GxRow cur;
    {
        const int q = blockIdx.x * 64 + (tid >> 2);
        int srow = 0, idx = -1;
        if (q < NP) {
            const int p = pair_point[q];
            srow = point_slot ? point_slot[p] : p;
            idx = nbr[(size_t)srow * k + (q - pair_off[p])];
        }
        cur = gx_fetch_row(idx, srow, tid & 3, x, pts, feat_geo);
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + PACKED_FLOATS);
        gx3 w_fw1 = frag + X3_FW1 + wave * (X3_T1 * 4 * 3 * 64) + lane;
        const WFragS fr1 = load_wfrags(w_fw1);                  // in flight during the gather
        T_MARK(31)
        // ---- gather: thread = (row, quarter of the 32-d latent); pieces straight into the planes.  The operands were requested
        //      during the previous tile (lookup chain pair -> point -> slot -> neighbour -> latent row).
        const int row0 = tid >> 2, q40 = tid & 3;
        {
            const int q = tile * 64 + row0;
            const float lo[4] = {cur.f0[0], cur.f0[1], cur.f0[2], cur.f0[3]}, hi[4] = {cur.f1[0], cur.f1[1], cur.f1[2], cur.f1[3]};
            store_quad_x3(X, row0, q40 * 8, lo);
            store_quad_x3(X, row0, q40 * 8 + 4, hi);
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (q40 == 0) {
                float d[4] = {cur.d[0], cur.d[1], cur.d[2], 0.f};
                if (cur.idx >= 0) {
                    const float dist = fmaxf(sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), 1e-12f);
                    const float sc = dist * rbf;
                    pair_tmp[(size_t)q * PT_STRIDE] = expf(-(sc * sc));
                }
                store_quad_x3(X, row0, 32, d);
                store_quad_x3(X, row0, 36, z);
            } else {                                             // K = 35 is padded to two k32-steps: columns 40 .. 63 are zero
                store_quad_x3(X, row0, 32 + 8 * q40, z);
                store_quad_x3(X, row0, 36 + 8 * q40, z);
            }
        }
        const int qn = (tile + (int)gridDim.x) * 64 + row0;       // this thread's row in the workgroup's next tile
        int n_p = qn < NP ? pair_point[qn] : -1, n_srow = 0, n_off = 0, n_idx = -1;
        T_MARK(0)
        lds_barrier();
        T_MARK(1)

        f32x4 acc[4][4];
        uint32_t m1[2], m2[2], m3[2], m4[2];
        float ssum[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- forward: 35 -> 256 -> 256 -> 256 -> 256 ---------------------------------------------------------------------------
        constexpr int WH = X3_TH * 4 * 3 * 64;                    // a wave's fragments of a hidden layer
        gx3 w_fw2 = frag + X3_FW2 + wave * WH + lane, w_fw3 = frag + X3_FW3 + wave * WH + lane;
        gx3 w_fw4 = frag + X3_FW4 + wave * WH + lane, w_bw4 = frag + X3_BW4 + wave * WH + lane;
        gx3 w_bw3 = frag + X3_BW3 + wave * WH + lane, w_bw2 = frag + X3_BW2 + wave * WH + lane;
        const int coff = 64 * wave + 4 * (lane >> 4);              // this lane's first feature of feature tile 0
        BiasS bias, v5q;
        WFragS nf;
        zero_acc(acc);
        gemm_xs<X3_T1, true, 1>(X, w_fw1, lane, acc, fr1, nf, w_fw2, pf + OFF_B1 + coff, pf, bias, v5q);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_xs<0, WITH_JAC>(X, acc, bias, bias, wave, lane, m1, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        zero_acc(acc);
        gemm_xs<X3_TH, true, 1>(X, w_fw2, lane, acc, nf, nf, w_fw3, pf + OFF_B2 + coff, pf, bias, v5q);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_xs<0, WITH_JAC>(X, acc, bias, bias, wave, lane, m2, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        if (n_p >= 0) n_idx = nbr[(size_t)n_srow * k + (qn - n_off)];
        zero_acc(acc);
        gemm_xs<X3_TH, true, 1>(X, w_fw3, lane, acc, nf, nf, w_fw4, pf + OFF_B3 + coff, pf, bias, v5q);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_xs<0, WITH_JAC>(X, acc, bias, bias, wave, lane, m3, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        cur = gx_fetch_row(n_idx, n_srow, q40, x, pts, feat_geo);
        zero_acc(acc);       // bias and the folded last layer v = T W8 (same quads) arrive with the last k-step
        gemm_xs<X3_TH, WITH_JAC, 2>(X, w_fw4, lane, acc, nf, nf, w_bw4, pf + OFF_B4 + coff, pf + OFF_V5 + coff, bias, v5q);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        // last forward layer: sdf_j = v . a4 + c from the accumulators; the planes receive the Jacobian seed v * lrelu'(h4)
        fwd_epilogue_xs<1, WITH_JAC>(X, acc, bias, v5q, wave, lane, m4, ssum);
        {
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float t = ssum[b] + __shfl_xor(ssum[b], 16);
                t += __shfl_xor(t, 32);
                if (g == 0) red[wave][16 * b + i] = t;
            }
        }
        T_MARK(6)
        lds_barrier();
        T_MARK(7)
        if (tid < 64) {
            const int q = tile * 64 + tid;
            if (q < NP) pair_tmp[(size_t)q * PT_STRIDE + 1] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + pf[OFF_C];
        }

        if (WITH_JAC) {
            // ---- Jacobian sweep: g_a3 = g_h4 W6 ; g_h3 = g_a3 * D3 ; ... ; J = g_h1 W0 ------------------------------------------
            zero_acc(acc);
            gemm_xs<X3_TH, true, 0>(X, w_bw4, lane, acc, nf, nf, w_bw3, pf, pf, bias, v5q);
            T_MARK(10)
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_xs(X, acc, wave, lane, m3);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
            zero_acc(acc);
            gemm_xs<X3_TH, true, 0>(X, w_bw3, lane, acc, nf, nf, w_bw2, pf, pf, bias, v5q);
            T_MARK(10)
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_xs(X, acc, wave, lane, m2);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
            zero_acc(acc);
            gemm_xs<X3_TH, false, 0>(X, w_bw2, lane, acc, nf, nf, w_bw2, pf, pf, bias, v5q);
            T_MARK(10)
            gx3 w_jw1 = frag + X3_JW1 + (wave >> 1) * (X3_TH * 2 * 3 * 64) + lane;
            const WFragS2 frj = load_wfrags2(w_jw1);
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_xs(X, acc, wave, lane, m1);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
            // last step 256 -> 35 (padded 64): wave = (feature half m, row half n), 2 x 2 tiles each
            {
                const int m = wave >> 1, n = wave & 1, i = lane & 15, g = lane >> 4;
                f32x4 aj[2][2];
                gemm_xs_tile<X3_TH>(X, n, w_jw1, lane, frj, aj);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int q = tile * 64 + 32 * n + 16 * b + i;
                    if (q < NP) {
                        if (m == 0) {
#pragma unroll
                            for (int a = 0; a < 2; ++a) *reinterpret_cast<f32x4*>(jac + (size_t)q * SPF_GEO_DIM + 16 * a + 4 * g) = aj[a][b];
                        } else if (g == 0) {
                            pair_tmp[(size_t)q * PT_STRIDE + 2] = aj[0][b][0];
                            pair_tmp[(size_t)q * PT_STRIDE + 3] = aj[0][b][1];
                            pair_tmp[(size_t)q * PT_STRIDE + 4] = aj[0][b][2];
                        }
                    }
                }
            }
        }
        T_MARK(14)
        lds_barrier();  // the planes are rewritten by the next tile's gather
        T_MARK(15)
    }
    T_FLUSH
    CLK_FLUSH(0, WITH_JAC ? 1 : 0)
}


// ==============================================================================================================================
// The same kernel on the 32 x 32 x 16 tile engine (mlp_tile_x3.h: 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 per wave) — the round-1
// form of the dominant kernel, kept selectable per call (arith = SPF_ARITH_SPLIT_W) so that bench.py can time BOTH MFMA shapes on the
// same batches in the same process on whatever box it runs on (roofline.ab): which shape holds the higher clock under load differs
// from box to box by about as much as the difference between them.  Same arithmetic (six exact bf16 piece products per fp32 product),
// own fragment image behind the 16 x 16 x 32 engine's inside the packed buffer.
// ==============================================================================================================================
constexpr int XW_T1 = 3;                          // first layer: K = 35 -> 48
constexpr int XW_TH = 16;                         // 256 / 16
constexpr int XW_SZ1 = 4 * XW_T1 * 2 * 3 * 64;    // bf16x8 entries
constexpr int XW_SZH = 4 * XW_TH * 2 * 3 * 64;
constexpr int XW_SZJ = 2 * XW_TH * 3 * 64;
constexpr int XW_FW1 = 0;
constexpr int XW_FW2 = XW_FW1 + XW_SZ1;
constexpr int XW_FW3 = XW_FW2 + XW_SZH;
constexpr int XW_FW4 = XW_FW3 + XW_SZH;
constexpr int XW_BW4 = XW_FW4 + XW_SZH;
constexpr int XW_BW3 = XW_BW4 + XW_SZH;
constexpr int XW_BW2 = XW_BW3 + XW_SZH;
constexpr int XW_JW1 = XW_BW2 + XW_SZH;
constexpr int XW_FRAGS = XW_JW1 + XW_SZJ;
constexpr int XW_BASE = PACKED_FLOATS + 4 * X3_FRAGS;          // float offset of this engine's fragments inside the packed image
constexpr int XH_BASE = XW_BASE + 4 * XW_FRAGS;                 // the same regions as fp16 piece pairs (H2 arithmetic): slot 2 of a triple unused
constexpr int PACKED_TOTAL = XH_BASE + 4 * XW_FRAGS;

// one thread per fragment slot (region, wave, k16, m, lane): 8 weights -> 3 x bf16x8, or (H2) 2 x f16x8 in the same slots
template <bool H2>
__global__ void geo_pack_x3w_kernel(PackArgs a, bf16x8* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int N1 = XW_SZ1 / 3, NH = XW_SZH / 3, NJ = XW_SZJ / 3;
    if (s >= N1 + 6 * NH + NJ) return;
    int region, local;
    if (s < N1) { region = 0; local = s; }
    else if (s < N1 + 6 * NH) { region = 1 + (s - N1) / NH; local = (s - N1) % NH; }
    else { region = 7; local = s - N1 - 6 * NH; }
    const int ln = local & 63, i = ln & 31, kg = ln >> 5;
    float w[8];
    size_t base;
    if (region < 7) {
        const int T = region == 0 ? XW_T1 : XW_TH;
        const int m = (local >> 6) & 1, t = (local >> 7) % T, wv = (local >> 7) / T;
        const int f = 64 * wv + 32 * m + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * t + 8 * kg + e;
            float v;
            switch (region) {
                case 0: v = k < K_IN ? a.w0[f * K_IN + k] : 0.f; break;
                case 1: v = a.w2[f * 256 + k]; break;
                case 2: v = a.w4[f * 256 + k]; break;
                case 3: v = a.w6[f * 256 + k]; break;
                case 4: v = a.w6[k * 256 + f]; break;      // g_a3[f] = sum_o g_h4[o] W6[o][f]
                case 5: v = a.w4[k * 256 + f]; break;
                default: v = a.w2[k * 256 + f]; break;
            }
            w[e] = v;
        }
        const int rb = region == 0 ? XW_FW1 : XW_FW2 + (region - 1) * XW_SZH;
        base = (size_t)rb + (size_t)((wv * T + t) * 2 + m) * 3 * 64 + ln;
    } else {
        const int t = (local >> 6) % XW_TH, m = (local >> 6) / XW_TH;
        const int f = 32 * m + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * t + 8 * kg + e;
            w[e] = f < K_IN ? a.w0[k * K_IN + f] : 0.f;  // J[f] = sum_o g_h1[o] W0[o][f]
        }
        base = (size_t)XW_JW1 + (size_t)(m * XW_TH + t) * 3 * 64 + ln;
    }
    if (H2) {
        f16x8 h1, h2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h1[e] = (_Float16)w[e];
            h2[e] = (_Float16)((w[e] - (float)h1[e]) * 2048.0f);
        }
        out[base] = __builtin_bit_cast(bf16x8, h1);
        out[base + 64] = __builtin_bit_cast(bf16x8, h2);
        out[base + 128] = bf16x8{};
        return;
    }
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 x, y, z;
        split3(w[e], x, y, z);
        p1[e] = x; p2[e] = y; p3[e] = z;
    }
    out[base] = p1;
    out[base + 64] = p2;
    out[base + 128] = p3;
}

// acc[m][n][4g + e] = feature 64w + 32m + 8g + 4kg + e of row 32n + j.  mask[m]: bit n * 16 + 4g + e.
// MODE 0: a = lrelu(acc + b) -> planes.   MODE 1 (last forward layer): also sdf partial sums s[n] += v5 . a, and the planes get
// the Jacobian seed v5 * lrelu'(h) instead of a (a itself is not needed any more).
// this lane's bias values of a layer (features 64 wave + 32 m + 8 g + 4 kg ..+3), requested ahead of the layer's GEMM
struct Bias3 {
    f32x4 b[2][4];
};
__device__ __forceinline__ Bias3 load_bias3(gfp bias, int wave, int lane) {
    Bias3 r;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            r.b[m][g] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(bias + 64 * wave + 32 * m + 8 * g + 4 * (lane >> 5));
    return r;
}

// forward epilogue: a = lrelu(acc + b) -> planes; sign bits pushed into mask[n] in the order (m, g, e) (32 per word).
// MODE 1 (last layer): sdf partial sums s[n] += v . a, and the planes receive the Jacobian seed v * lrelu'(h) instead.
// (round 5 tried fewer vector instructions here — v_max LeakyReLU, sign bits four at a time from the top pieces, the bias as the first product's
// C operand: a measured null result, profiles/r05_epilogue_ab.json; removed in round 6, the record stays)
template <int MODE, bool WITH_JAC, int NT = 2, bool H2 = false>
__device__ __forceinline__ void fwd_epilogue_x3(__bf16* X, const f32x16 (&acc)[2][NT], const Bias3& bias, const Bias3& v5, int wave, int lane,
                                                uint32_t (&mask)[NT], float (&s)[NT]) {
    const int j = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int n = 0; n < NT; ++n) mask[n] = 0u;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
            const f32x4 bv = bias.b[m][g];
            f32x4 vv = f32x4{0.f, 0.f, 0.f, 0.f}, vs = vv;
            if (MODE == 1) {
                vv = v5.b[m][g];
                vs = vv * 0.01f;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 h, hs;
                bias_scale4(acc[m][n], g, bv, h, hs);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (MODE == 1) {
                        float seed;
                        const float a = lrelu_push_sel(h[e], hs[e], vv[e], vs[e], seed, mask[n]);
                        s[n] += vv[e] * a;
                        out[e] = seed;
                    } else {
                        out[e] = lrelu_push(h[e], hs[e], mask[n]);
                    }
                }
                if (MODE == 0 || WITH_JAC) store_quad_xh<H2, X3_LDP>(X, 32 * n + j, f0, out);
            }
        }
}

// backward epilogue: g_h = g_a * lrelu'(h) -> planes (pops the words the forward epilogue filled, in the same order)
template <int NT = 2, bool H2 = false>
__device__ __forceinline__ void bwd_epilogue_x3(__bf16* X, const f32x16 (&acc)[2][NT], int wave, int lane, const uint32_t (&mask_in)[NT]) {
    const int j = lane & 31, kg = lane >> 5;
    uint32_t mask[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) mask[n] = mask_in[n];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 v, vs;
                scale4(acc[m][n], g, v, vs);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = lrelu_pop(v[e], vs[e], mask[n]);
                store_quad_xh<H2, X3_LDP>(X, 32 * n + j, f0, out);
            }
        }
}


// One workgroup's tiles.  NT = 2: tiles of 64 consecutive valid pairs (the round-1..4 kernel, unchanged); NT = 1 (round 5): HALF-HEIGHT tiles of 32
// pairs — the weights stream as for a 64-row tile (the k-step becomes L1-fill-bound instead of matrix-pipe-bound: mlp_tile_x3.h), so a tile
// takes ~55 % of a full one's time and twice as many workgroups have work: chosen by the kernel when all the launch's pairs fit ONE pass of
// half tiles over the grid (the sampler pass and the pseudo-point pass of a 128-ray step: 64 resp. 16 full tiles on a 256-CU chip).
// a layer's GEMM on the engine the body was built for: bf16 pieces (one accumulator), or H2 (main + cross accumulators, combined here)
template <int T, int NT, int NPC, bool H2, int NC>
__device__ __forceinline__ WFrag3 gemm_xh(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][NT], const WFrag3& first, gx3 next_wp,
                                          f32x16 (&accc)[2][NC]) {
    if constexpr (H2) {
        static_assert(NC == NT, "H2: one cross accumulator per main accumulator");
        const WFrag3 nf = gemm_x3<T, false, X3_LDP, NT, 2, true>(X, wp, lane, acc, first, next_wp, accc);
        h2_combine<NT>(acc, accc);
        return nf;
    } else {
        return gemm_x3<T, false, X3_LDP, NT, NPC>(X, wp, lane, acc, first, next_wp);
    }
}

template <bool WITH_JAC, int NT, int NPC = 3, bool H2 = false>
__device__ __forceinline__ void geo_x3w_body(__bf16* X, float (*red)[64], const float* __restrict__ x, const int32_t* __restrict__ nbr,
                                             const int32_t* __restrict__ point_slot, const int32_t* __restrict__ pair_off,
                                             const int32_t* __restrict__ pair_point, const int NP, const int q0, int k, const float* __restrict__ pts,
                                             const float* __restrict__ feat_geo, const float* packed, float rbf, float* __restrict__ pair_tmp,
                                             float* __restrict__ jac) {
    constexpr int ROWS = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (NP + ROWS - 1) / ROWS;
    const float* packed0 = packed;
    T_DECL
    GxRow cur;
    {
        const int q = blockIdx.x * ROWS + (tid >> 2);
        int srow = 0, idx = -1;
        if ((tid >> 2) < ROWS && q < NP) {
            const int p = pair_point[q];
            srow = point_slot ? point_slot[p] : p;
            idx = nbr[(size_t)srow * k + (q + q0 - pair_off[p])];
        }
        cur = gx_fetch_row(idx, srow, tid & 3, x, pts, feat_geo);
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        gfp pf = launder(packed0);
        gx3 frag = reinterpret_cast<gx3>(pf + (H2 ? XH_BASE : XW_BASE));
        gx3 w_fw1 = frag + XW_FW1 + wave * (XW_T1 * 2 * 3 * 64) + lane;
        const WFrag3 fr1 = load_wfrag3(w_fw1);                  // in flight during the gather
        T_MARK(31)
        // ---- gather: thread = (row, quarter of the 32-d latent); pieces straight into the planes.  The operands were requested
        //      during the previous tile (lookup chain pair -> point -> slot -> neighbour -> latent row).
        const int row0 = tid >> 2, q40 = tid & 3;
        if (row0 < ROWS) {
            const int q = tile * ROWS + row0;
            const float lo[4] = {cur.f0[0], cur.f0[1], cur.f0[2], cur.f0[3]}, hi[4] = {cur.f1[0], cur.f1[1], cur.f1[2], cur.f1[3]};
            store_quad_xh<H2, X3_LDP>(X, row0, q40 * 8, lo);
            store_quad_xh<H2, X3_LDP>(X, row0, q40 * 8 + 4, hi);
            if (q40 == 0) {
                float d[4] = {cur.d[0], cur.d[1], cur.d[2], 0.f};
                if (cur.idx >= 0) {
                    const float dist = fmaxf(sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), 1e-12f);
                    const float sc = dist * rbf;
                    pair_tmp[(size_t)q * PT_STRIDE] = expf(-(sc * sc));
                }
                store_quad_xh<H2, X3_LDP>(X, row0, 32, d);
                const float z[4] = {0.f, 0.f, 0.f, 0.f};
                store_quad_xh<H2, X3_LDP>(X, row0, 36, z);
                store_quad_xh<H2, X3_LDP>(X, row0, 40, z);
                store_quad_xh<H2, X3_LDP>(X, row0, 44, z);
            }
        }
        const int qn = (tile + (int)gridDim.x) * ROWS + row0;     // this thread's row in the workgroup's next tile
        int n_p = (row0 < ROWS && qn < NP) ? pair_point[qn] : -1, n_srow = 0, n_off = 0, n_idx = -1;
        T_MARK(0)
        lds_barrier();
        T_MARK(1)

        f32x16 acc[2][NT];
        f32x16 accc[2][H2 ? NT : 1];          // H2: the cross terms' accumulators
        uint32_t m1[NT], m2[NT], m3[NT], m4[NT];
        float ssum[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) ssum[n] = 0.f;
        // ---- forward: 35 -> 256 -> 256 -> 256 -> 256 ---------------------------------------------------------------------------
        gx3 w_fw2 = frag + XW_FW2 + wave * (XW_TH * 2 * 3 * 64) + lane, w_fw3 = frag + XW_FW3 + wave * (XW_TH * 2 * 3 * 64) + lane;
        gx3 w_fw4 = frag + XW_FW4 + wave * (XW_TH * 2 * 3 * 64) + lane, w_bw4 = frag + XW_BW4 + wave * (XW_TH * 2 * 3 * 64) + lane;
        gx3 w_bw3 = frag + XW_BW3 + wave * (XW_TH * 2 * 3 * 64) + lane, w_bw2 = frag + XW_BW2 + wave * (XW_TH * 2 * 3 * 64) + lane;
        Bias3 bias = load_bias3(pf + OFF_B1, wave, lane);
        WFrag3 nf;
        nf = gemm_xh<XW_T1, NT, NPC, H2>(X, w_fw1, lane, acc, fr1, w_fw2, accc);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_x3<0, WITH_JAC, NT, H2>(X, acc, bias, bias, wave, lane, m1, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        if (n_p >= 0) {
            n_srow = point_slot ? point_slot[n_p] : n_p;
            n_off = pair_off[n_p];
        }
        bias = load_bias3(pf + OFF_B2, wave, lane);
        nf = gemm_xh<XW_TH, NT, NPC, H2>(X, w_fw2, lane, acc, nf, w_fw3, accc);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_x3<0, WITH_JAC, NT, H2>(X, acc, bias, bias, wave, lane, m2, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        if (n_p >= 0) n_idx = nbr[(size_t)n_srow * k + (qn + q0 - n_off)];
        bias = load_bias3(pf + OFF_B3, wave, lane);
        nf = gemm_xh<XW_TH, NT, NPC, H2>(X, w_fw3, lane, acc, nf, w_fw4, accc);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        fwd_epilogue_x3<0, WITH_JAC, NT, H2>(X, acc, bias, bias, wave, lane, m3, ssum);
        T_MARK(4)
        lds_barrier();
        T_MARK(5)
        cur = gx_fetch_row(n_idx, n_srow, q40, x, pts, feat_geo);
        bias = load_bias3(pf + OFF_B4, wave, lane);
        const Bias3 v5q = load_bias3(pf + OFF_V5, wave, lane);       // folded last layer v = T W8, same quads: requested ahead of the GEMM too
        nf = gemm_xh<XW_TH, NT, NPC, H2>(X, w_fw4, lane, acc, nf, WITH_JAC ? w_bw4 : nullptr, accc);
        T_MARK(2)
        lds_barrier();
        T_MARK(3)
        // last forward layer: sdf_j = v . a4 + c from the accumulators; the planes receive the Jacobian seed v * lrelu'(h4)
        fwd_epilogue_x3<1, WITH_JAC, NT, H2>(X, acc, bias, v5q, wave, lane, m4, ssum);
        {
            const int j = lane & 31, kg = lane >> 5;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float t = ssum[n] + __shfl_xor(ssum[n], 32);
                if (kg == 0) red[wave][32 * n + j] = t;
            }
        }
        T_MARK(6)
        lds_barrier();
        T_MARK(7)
        if (tid < ROWS) {
            const int q = tile * ROWS + tid;
            if (q < NP) pair_tmp[(size_t)q * PT_STRIDE + 1] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + pf[OFF_C];
        }

        if (WITH_JAC) {
            // ---- Jacobian sweep: g_a3 = g_h4 W6 ; g_h3 = g_a3 * D3 ; ... ; J = g_h1 W0 ------------------------------------------
                        nf = gemm_xh<XW_TH, NT, (H2 ? 2 : 3), H2>(X, w_bw4, lane, acc, nf, w_bw3, accc);
            T_MARK(10)
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_x3<NT, H2>(X, acc, wave, lane, m3);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
                        nf = gemm_xh<XW_TH, NT, (H2 ? 2 : 3), H2>(X, w_bw3, lane, acc, nf, w_bw2, accc);
            T_MARK(10)
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_x3<NT, H2>(X, acc, wave, lane, m2);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
                        gemm_xh<XW_TH, NT, (H2 ? 2 : 3), H2>(X, w_bw2, lane, acc, nf, nullptr, accc);
            T_MARK(10)
            gx3 w_jw1 = frag + XW_JW1 + (wave >> 1) * (XW_TH * 3 * 64) + lane;
            const WFrag1 frj = load_wfrag1(w_jw1);
            lds_barrier();
            T_MARK(11)
            bwd_epilogue_x3<NT, H2>(X, acc, wave, lane, m1);
            T_MARK(12)
            lds_barrier();
            T_MARK(13)
            // last step 256 -> 35 (padded 64): wave = (feature half m, row half n), one 32x32 tile each
            if (NT == 2 || (wave & 1) == 0) {                   // (half tiles: the two waves of the second row half have nothing to multiply)
                const int m = wave >> 1, n = wave & 1, j = lane & 31, kg = lane >> 5;
                const f32x16 aj = gemm_x3_tile<XW_TH, X3_LDP, H2>(X, n, w_jw1, lane, frj);
                const int q = tile * ROWS + 32 * n + j;
                if (q < NP) {
                    if (m == 0) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<f32x4*>(jac + (size_t)q * SPF_GEO_DIM + 8 * g + 4 * kg) = f32x4{aj[4 * g], aj[4 * g + 1], aj[4 * g + 2], aj[4 * g + 3]};
                    } else if (kg == 0) {
                        pair_tmp[(size_t)q * PT_STRIDE + 2] = aj[0];
                        pair_tmp[(size_t)q * PT_STRIDE + 3] = aj[1];
                        pair_tmp[(size_t)q * PT_STRIDE + 4] = aj[2];
                    }
                }
            }
        }
        T_MARK(14)
        lds_barrier();  // the planes are rewritten by the next tile's gather
        T_MARK(15)
    }
    T_FLUSH
}

template <bool WITH_JAC, int NPC = 3, bool H2 = false>
__global__ void __launch_bounds__(256, 1)
geo_pairs_x3w_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot,
                    const int32_t* __restrict__ pair_off, const int32_t* __restrict__ pair_point, const int32_t* __restrict__ n_pairs_dev,
                    int max_pairs, int k, const float* __restrict__ pts, const float* __restrict__ feat_geo, const float* packed,
                    float rbf, float* __restrict__ pair_tmp, float* __restrict__ jac, int clk) {
    __shared__ __attribute__((aligned(16))) __bf16 X[X3_LDS_BF16];
    __shared__ float red[4][64];
    const int NP = n_pairs_dev ? min(*n_pairs_dev, max_pairs) : max_pairs;
    CLK_DECL
    // Whole rounds of 64-pair tiles first (every workgroup one tile per round); what is left — less than one round — as HALF-HEIGHT tiles
    // when that is at most one per workgroup (a half tile takes ~55 % of a full one's time), else as full tiles.  A 128-ray step's main pass
    // (49 k pairs = 770 tiles on 256 workgroups) was 3.01 rounds paid as 4; it is 3 rounds + one half-tile pass now.  The body sees the
    // remainder through shifted pair arrays (q0 = its first pair: only the neighbour-column lookup needs the absolute pair number).
    const int G = (int)gridDim.x;
    const int q_full = (NP / (64 * G)) * (64 * G), rem = NP - q_full;
    if (q_full > 0)
        geo_x3w_body<WITH_JAC, 2, NPC, H2>(X, red, x, nbr, point_slot, pair_off, pair_point, q_full, 0, k, pts, feat_geo, packed, rbf, pair_tmp, jac);
    if (rem > 0) {
        const int32_t* pp = pair_point + q_full;
        float* pt = pair_tmp + (size_t)q_full * PT_STRIDE;
        float* jc = jac ? jac + (size_t)q_full * SPF_GEO_DIM : nullptr;
        if (q_full > 0) lds_barrier();          // (the first body's last tile is done with the planes)
        if (rem <= 32 * G)
            geo_x3w_body<WITH_JAC, 1, NPC, H2>(X, red, x, nbr, point_slot, pair_off, pp, rem, q_full, k, pts, feat_geo, packed, rbf, pt, jc);
        else
            geo_x3w_body<WITH_JAC, 2, NPC, H2>(X, red, x, nbr, point_slot, pair_off, pp, rem, q_full, k, pts, feat_geo, packed, rbf, pt, jc);
    }
    CLK_FLUSH(1, WITH_JAC ? 1 : 0)
}

}  // namespace

SPF_DEFINE_TIMING_ENTRY(spf_debug_timing_geo)

extern "C" {

int64_t spf_geo_packed_floats(void) { return PACKED_TOTAL; }

int spf_geo_clock_read(uint64_t* out12, int32_t reset) {
    if (!out12) return spf::fail(SPF_EINVAL, "spf_geo_clock_read: null pointer");
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(spf_geo_clock_buf), 12 * sizeof(uint64_t)) != hipSuccess)
        return spf::fail(SPF_EHIP, "spf_geo_clock_read: hipMemcpyFromSymbol failed");
    if (reset) {
        const uint64_t z[12] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(spf_geo_clock_buf), z, sizeof(z)) != hipSuccess)
            return spf::fail(SPF_EHIP, "spf_geo_clock_read: hipMemcpyToSymbol failed");
    }
    return SPF_OK;
}

int spf_geo_pack(const float* w0, const float* b0, const float* w2, const float* b2, const float* w4, const float* b4,
                 const float* w6, const float* b6, const float* w8, const float* b8, const float* wT, const float* bT,
                 float* packed, void* stream) {
    if (!w0 || !b0 || !w2 || !b2 || !w4 || !b4 || !w6 || !b6 || !w8 || !b8 || !wT || !bT || !packed)
        return spf::fail(SPF_EINVAL, "spf_geo_pack: null pointer");
    PackArgs a{w0, b0, w2, b2, w4, b4, w6, b6, w8, b8, wT, bT};
    geo_pack_kernel<<<spf::div_up(PACKED_FLOATS, 256), 256, 0, (hipStream_t)stream>>>(a, packed);
    geo_pack_x3_kernel<<<spf::div_up(X3_FRAGS / 3, 256), 256, 0, (hipStream_t)stream>>>(a, reinterpret_cast<bf16x8*>(packed + PACKED_FLOATS));
    geo_pack_x3w_kernel<false><<<spf::div_up(XW_FRAGS / 3, 256), 256, 0, (hipStream_t)stream>>>(a, reinterpret_cast<bf16x8*>(packed + XW_BASE));
    geo_pack_x3w_kernel<true><<<spf::div_up(XW_FRAGS / 3, 256), 256, 0, (hipStream_t)stream>>>(a, reinterpret_cast<bf16x8*>(packed + XH_BASE));
    SPF_LAUNCH_CHECK("geo_pack_kernel");
    return SPF_OK;
}

int spf_geo_forward(const float* x, const int32_t* nbr, const int32_t* point_slot, const int32_t* pair_off, const int32_t* pair_point,
                    const int32_t* n_points, const int32_t* n_pairs, int32_t max_points, int32_t max_pairs, int32_t k, const float* pts,
                    const float* feat_geo, const float* packed, float rbf, float* sdf, float* grad, float* wn, float* jac,
                    float* pair_tmp, int32_t arith, void* stream) {
    const int clk = (arith & SPF_ARITH_CLOCK) ? 1 : 0;      // opt-in held-clock stamps (spf_geo_clock_read); ignored by the fp32-MFMA twin
    const bool lite = (arith & SPF_ARITH_LITE) != 0;        // reduced products (two pieces per operand): SDF-only passes of the 32x32x16 engine
    arith &= ~(SPF_ARITH_CLOCK | SPF_ARITH_LITE);
    if (lite && arith == SPF_ARITH_H2) {      // (H2 already takes three products per fp32 product: nothing to reduce)
    } else if (lite && (arith != SPF_ARITH_SPLIT_W || grad || jac))
        return spf::fail(SPF_EINVAL, "spf_geo_forward: SPF_ARITH_LITE goes with SPF_ARITH_SPLIT_W and an SDF-only pass (no grad / jac): its values steer the sampler, nothing else");
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32 && arith != SPF_ARITH_SPLIT_W && arith != SPF_ARITH_H2)
        return spf::fail(SPF_EINVAL, "spf_geo_forward: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1), SPF_ARITH_SPLIT_W (2) or SPF_ARITH_H2 (3) [| SPF_ARITH_CLOCK], got %d", arith);
    if (max_points < 0 || max_pairs < 0 || k < 1 || k > SPF_KMAX)
        return spf::fail(SPF_EINVAL, "spf_geo_forward: bad sizes (max_points=%d max_pairs=%d k=%d)", max_points, max_pairs, k);
    if (max_points == 0 || max_pairs == 0) return SPF_OK;
    if (!x || !nbr || !pair_off || !pair_point || !pts || !feat_geo || !packed || !pair_tmp)
        return spf::fail(SPF_EINVAL, "spf_geo_forward: null pointer");
    if (!sdf && grad) return spf::fail(SPF_EINVAL, "spf_geo_forward: sdf may only be left out (per-pair scratch for the caller's own reduction) without grad / jac");
    if ((grad == nullptr) != (jac == nullptr)) return spf::fail(SPF_EINVAL, "spf_geo_forward: grad and jac must be given together");
    if (grad && !wn) return spf::fail(SPF_EINVAL, "spf_geo_forward: wn is required with grad/jac");
    const int tiles = spf::div_up(max_pairs, 64);
    hipStream_t s = (hipStream_t)stream;
    const int blocks = tiles < 512 ? tiles : 512;  // 2 workgroups per CU x 256 CUs, tiles are strided over them
    if (arith == SPF_ARITH_SPLIT) {
        const int b1 = tiles < 256 ? tiles : 256;   // one workgroup per CU (the bf16 planes take 101 KB of LDS)
        if (grad)
            geo_pairs_x3_kernel<true><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed, rbf,
                                                         pair_tmp, jac, clk);
        else
            geo_pairs_x3_kernel<false><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed,
                                                          rbf, pair_tmp, nullptr, clk);
    } else if (arith == SPF_ARITH_H2) {
        const int half = spf::div_up(max_pairs, 32);
        const int b1 = half < 256 ? half : 256;
        if (grad)
            geo_pairs_x3w_kernel<true, 2, true><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed, rbf,
                                                                   pair_tmp, jac, clk);
        else
            geo_pairs_x3w_kernel<false, 2, true><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed,
                                                                    rbf, pair_tmp, nullptr, clk);
    } else if (arith == SPF_ARITH_SPLIT_W) {
        const int half = spf::div_up(max_pairs, 32);          // the kernel takes half-height (32-pair) tiles when they all fit one pass
        const int b1 = half < 256 ? half : 256;
        if (grad)
            geo_pairs_x3w_kernel<true><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed, rbf,
                                                          pair_tmp, jac, clk);
        else if (lite)
            geo_pairs_x3w_kernel<false, 2><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed,
                                                              rbf, pair_tmp, nullptr, clk);
        else
            geo_pairs_x3w_kernel<false><<<b1, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed,
                                                           rbf, pair_tmp, nullptr, clk);
    } else if (grad)
        geo_pairs_kernel<true><<<blocks, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed, rbf,
                                                      pair_tmp, jac);
    else
        geo_pairs_kernel<false><<<blocks, 256, 0, s>>>(x, nbr, point_slot, pair_off, pair_point, n_pairs, max_pairs, k, pts, feat_geo, packed,
                                                       rbf, pair_tmp, nullptr);
    SPF_LAUNCH_CHECK("geo_pairs_kernel");
    if (!sdf) return SPF_OK;      // ABI 5: pair_tmp = [w, sdf_j, ...] per pair is the result; spf_sampler_train forms the per-point means on its way
    geo_point_reduce_kernel<<<spf::div_up(max_points, 256), 256, 0, s>>>(pair_tmp, pair_off, point_slot, n_points, max_points, sdf, grad, wn);
    SPF_LAUNCH_CHECK("geo_point_reduce_kernel");
    return SPF_OK;
}

int spf_geo_backward_latents(const float* g_sdf, const float* wn, const float* jac, const int32_t* nbr, const int32_t* point_slot,
                             const int32_t* pair_off, const int32_t* pair_point, const int32_t* n_pairs, int32_t max_pairs, int32_t k,
                             float* g_feat_geo, int64_t* g_feat_geo_fixed, const float* grad_x, float* g_x, int32_t n_rows, void* stream) {
    if (max_pairs < 0 || k < 1 || k > SPF_KMAX || n_rows < 0) return spf::fail(SPF_EINVAL, "spf_geo_backward_latents: bad sizes");
    if (g_x && !grad_x) return spf::fail(SPF_EINVAL, "spf_geo_backward_latents: g_x needs grad_x");
    if (max_pairs == 0 && !(g_x && n_rows > 0)) return SPF_OK;
    if (!g_sdf || !wn || !jac || !nbr || !pair_off || !pair_point || (!g_feat_geo && !g_feat_geo_fixed))
        return spf::fail(SPF_EINVAL, "spf_geo_backward_latents: null pointer");
    int blocks = spf::div_up(max_pairs, 64);                              // one workgroup per 64-pair tile, grid-strided
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    geo_backward_latents_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g_sdf, wn, jac, nbr, point_slot, pair_off, pair_point, n_pairs,
                                                                         max_pairs, k, g_feat_geo, reinterpret_cast<long long*>(g_feat_geo_fixed),
                                                                         grad_x, g_x, n_rows);
    SPF_LAUNCH_CHECK("geo_backward_latents_kernel");
    return SPF_OK;
}

}  // extern "C"
