// Weight packing of the colour trunk (spf_color_pack) as device functions + the packed image's layout constants: shared by color_mlp.hip and the
// step prologue launch (camera.hip: spf_step_prologue), in which the packing rides since round 5 (it only reads the parameters).
#pragma once
#include "mlp_tile.h"
#include "mlp_tile_x3.h"

namespace {

using namespace spf;

constexpr int C_IN = 103;   // 39 posenc + 64 latent (pointneus_disent.py:329-330: [posenc | feat]).
                            // Inside the kernels the columns are permuted to [latent(64) | posenc(39) | pad] so the latent
                            // gather lands 16-B aligned in LDS; spf_color_pack applies the permutation to W0's columns,
                            // and act0 / dW0 are in this internal order (internal k -> reference column: c_orig()).
constexpr int C_INP = 104;  // padded to a multiple of 8
constexpr int T_CIN = 13;
constexpr int N_FREQ = 6;   // get_embedder(multires=6), pointneus_disent.py:70-72

constexpr int SZ_CFW1 = 4 * T_CIN * 2 * 64 * 4;
constexpr int SZ_CHH = 4 * T_HID * 2 * 64 * 4;
constexpr int SZ_CBL = 2 * T_HID * 64 * 4;
constexpr int CO_FW1 = 0;
constexpr int CO_FW2 = CO_FW1 + SZ_CFW1;
constexpr int CO_FW3 = CO_FW2 + SZ_CHH;
constexpr int CO_BW3 = CO_FW3 + SZ_CHH;
constexpr int CO_BW2 = CO_BW3 + SZ_CHH;
constexpr int CO_BWL = CO_BW2 + SZ_CHH;   // W0[:, 39:103] (256 -> 64 latent columns)
constexpr int CO_B1 = CO_BWL + SZ_CBL;
constexpr int CO_B2 = CO_B1 + 256;
constexpr int CO_B3 = CO_B2 + 256;
constexpr int C_PACKED = CO_B3 + 256;

constexpr int CL_X = 0;
constexpr int CL_W = CL_X + 64 * LDA;   // forward: per row {normalised RBF weight, compact point id bits (-1 = padding)}; backward: neighbour ids
constexpr int CL_TOTAL = CL_W + 128;

// ---- pack ------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int c_orig(int k) { return k < 64 ? 39 + k : k - 64; }

struct CPackArgs {
    const float *w0, *b0, *w2, *b2, *w4, *b4;
};

__device__ __forceinline__ void color_pack_kernel_body(const CPackArgs& a, float* __restrict__ out, int e) {
    if (e >= C_PACKED) return;
    float val = 0.f;
    if (e < CO_BWL) {
        int region, local;
        if (e < CO_FW2) { region = 0; local = e; }
        else { region = 1 + (e - CO_FW2) / SZ_CHH; local = (e - CO_FW2) % SZ_CHH; }
        const int T = region == 0 ? T_CIN : T_HID;
        const int j = local & 3, ln = (local >> 2) & 63, nt = (local >> 8) & 1;
        const int t = (local >> 9) % T, w = (local >> 9) / T;
        const int n = 64 * w + 32 * nt + (ln & 31), kk = 8 * t + 4 * (ln >> 5) + j;
        switch (region) {
            case 0: val = kk < C_IN ? a.w0[n * C_IN + c_orig(kk)] : 0.f; break;
            case 1: val = a.w2[n * 256 + kk]; break;
            case 2: val = a.w4[n * 256 + kk]; break;
            case 3: val = a.w4[kk * 256 + n]; break;   // g_a2[i] = sum_o G3[o] W4[o][i]
            default: val = a.w2[kk * 256 + n]; break;
        }
    } else if (e < CO_B1) {
        const int local = e - CO_BWL;
        const int j = local & 3, ln = (local >> 2) & 63, t = (local >> 8) & 31, nt = (local >> 13) & 1;
        const int n = 39 + 32 * nt + (ln & 31), kk = 8 * t + 4 * (ln >> 5) + j;
        val = a.w0[kk * C_IN + n];
    } else {
        const int local = e - CO_B1, l = local >> 8, i = local & 255;
        val = (l == 0 ? a.b0 : l == 1 ? a.b2 : a.b4)[i];
    }
    out[e] = val;
}


// ---- the bf16-piece (x3) engine's fragment image (layout: color_mlp.hip, "The same two kernels on the bf16 matrix pipe") ----
constexpr int CX_T1 = 7;                          // layer 0: K = 104 -> 112
constexpr int CX_TH = 16;
constexpr int CX_SZ1 = 4 * CX_T1 * 2 * 3 * 64;    // bf16x8 entries
constexpr int CX_SZH = 4 * CX_TH * 2 * 3 * 64;
constexpr int CX_SZL = 2 * CX_TH * 3 * 64;
constexpr int CX_FW1 = 0;
constexpr int CX_FW2 = CX_FW1 + CX_SZ1;
constexpr int CX_FW3 = CX_FW2 + CX_SZH;
constexpr int CX_BW3 = CX_FW3 + CX_SZH;
constexpr int CX_BW2 = CX_BW3 + CX_SZH;
constexpr int CX_BWL = CX_BW2 + CX_SZH;
constexpr int CX_FRAGS = CX_BWL + CX_SZL;
constexpr int CH_OFF = C_PACKED + 4 * CX_FRAGS;                  // float offset of the SAME fragment image as fp16 piece pairs (H2 arithmetic; slot 2 unused)
constexpr int C_PACKED_TOTAL = CH_OFF + 4 * CX_FRAGS;

__device__ __forceinline__ void color_pack_x3_kernel_body(const CPackArgs& a, bf16x8* __restrict__ out, int s) {
    constexpr int N1 = CX_SZ1 / 3, NH = CX_SZH / 3, NL = CX_SZL / 3;
    if (s >= N1 + 4 * NH + NL) return;
    int region, local;
    if (s < N1) { region = 0; local = s; }
    else if (s < N1 + 4 * NH) { region = 1 + (s - N1) / NH; local = (s - N1) % NH; }
    else { region = 5; local = s - N1 - 4 * NH; }
    const int ln = local & 63, i = ln & 31, kg = ln >> 5;
    float w[8];
    size_t base;
    if (region < 5) {
        const int T = region == 0 ? CX_T1 : CX_TH;
        const int m = (local >> 6) & 1, t = (local >> 7) % T, wv = (local >> 7) / T;
        const int f = 64 * wv + 32 * m + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * t + 8 * kg + e;
            float v;
            switch (region) {
                case 0: v = k < C_IN ? a.w0[f * C_IN + c_orig(k)] : 0.f; break;
                case 1: v = a.w2[f * 256 + k]; break;
                case 2: v = a.w4[f * 256 + k]; break;
                case 3: v = a.w4[k * 256 + f]; break;      // g_a2[f] = sum_o G3[o] W4[o][f]
                default: v = a.w2[k * 256 + f]; break;
            }
            w[e] = v;
        }
        const int rb = region == 0 ? CX_FW1 : CX_FW2 + (region - 1) * CX_SZH;
        base = (size_t)rb + (size_t)((wv * T + t) * 2 + m) * 3 * 64 + ln;
    } else {
        const int t = (local >> 6) % CX_TH, m = (local >> 6) / CX_TH;
        const int f = 32 * m + i;                         // latent column 0..63 = reference column 39 + f
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = a.w0[(16 * t + 8 * kg + e) * C_IN + 39 + f];
        base = (size_t)CX_BWL + (size_t)(m * CX_TH + t) * 3 * 64 + ln;
    }
    bf16x8 p1, p2, p3;
    f16x8 h1, h2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 x, y, z;
        split3(w[e], x, y, z);
        p1[e] = x; p2[e] = y; p3[e] = z;
        h1[e] = (_Float16)w[e];
        h2[e] = (_Float16)((w[e] - (float)h1[e]) * 2048.0f);
    }
    out[base] = p1;
    out[base + 64] = p2;
    out[base + 128] = p3;
    out[CX_FRAGS + base] = __builtin_bit_cast(bf16x8, h1);          // the H2 image: the same slots, two fp16 pieces
    out[CX_FRAGS + base + 64] = __builtin_bit_cast(bf16x8, h2);
}

// the whole packed image (fp32 image + bf16-piece fragments) + an optional buffer to clear, for thread e of `nthreads`
constexpr int C_PACK_THREADS = C_PACKED > CX_FRAGS / 3 ? C_PACKED : CX_FRAGS / 3;
__device__ __forceinline__ void color_pack_all(const CPackArgs& a, float* __restrict__ out, float* __restrict__ zero_buf, long long zero_floats, long long e,
                                               long long nthreads) {
    if (e < C_PACK_THREADS) {
        color_pack_kernel_body(a, out, (int)e);
        color_pack_x3_kernel_body(a, reinterpret_cast<bf16x8*>(out + C_PACKED), (int)e);
    }
    // optional: clear a buffer the following forward accumulates into (atomics) / scatters into — its fill launch rides along
    if (zero_buf) {
        f32x4* z4 = reinterpret_cast<f32x4*>(zero_buf);
        const long long n4 = zero_floats >> 2;
        for (long long i = e; i < n4; i += nthreads) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < (zero_floats & 3)) zero_buf[(n4 << 2) + e] = 0.f;
    }
}

}  // namespace
