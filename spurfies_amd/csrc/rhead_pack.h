// Weight packing of the per-point head stage (spf_rhead_pack) as device functions + the packed image's layout constants: shared by rhead_mlp.hip
// and the step prologue launch (camera.hip: spf_step_prologue), in which the packing rides since round 5 (it only reads the parameters).
#pragma once
#include "mlp_tile.h"
#include "mlp_tile_x3.h"

namespace {

using namespace spf;

constexpr int R_IN = 277;      // 21 dir-enc + 256 agg (pointneus_disent.py:343-344: [encoded_dir | agg])
constexpr int T_RIN = 35;
constexpr int LDR = 284;       // LDS row stride (1136 B): 284 mod 64 = 28 -> ds_read_b128 conflict-free
constexpr int DIR_FREQ = 3;    // get_embedder(multires=3), pointneus_disent.py:73-75

constexpr int SZ_RFW1 = 4 * T_RIN * 2 * 64 * 4;
constexpr int SZ_RHH = 4 * T_HID * 2 * 64 * 4;
constexpr int RO_FW1 = 0;
constexpr int RO_FW2 = RO_FW1 + SZ_RFW1;
constexpr int RO_BW2 = RO_FW2 + SZ_RHH;    // g_a1 = G2 * W2
constexpr int RO_BWA = RO_BW2 + SZ_RHH;    // g_agg = G1 * W1[:, 21:277]
constexpr int RO_FW6 = RO_BWA + SZ_RHH;    // agg = agg3 * W6^T      (F_color.6)
constexpr int RO_BW6 = RO_FW6 + SZ_RHH;    // g_agg3 = g_agg * W6
constexpr int RO_B1 = RO_BW6 + SZ_RHH;
constexpr int RO_B2 = RO_B1 + 256;
constexpr int RO_B6 = RO_B2 + 256;
constexpr int RO_W3 = RO_B6 + 256;         // [3][256]
constexpr int RO_B3 = RO_W3 + 768;         // [3] (+1 pad)
constexpr int R_PACKED = RO_B3 + 4;

constexpr int RL_X = 0;
constexpr int RL_G3 = RL_X + 64 * LDR;     // [64][4] per-row dL/d(pre-sigmoid)
constexpr int RL_ROW = RL_G3 + 256;        // [64] slot row of each point (int bits), -1 = padding
constexpr int RL_TOTAL = RL_ROW + 64;

__host__ __device__ __forceinline__ int r_orig(int k) { return k < 256 ? 21 + k : k - 256; }  // internal column -> reference column

struct RPackArgs {
    const float *w6, *b6, *w0, *b0, *w2, *b2, *w4, *b4;
};

__device__ __forceinline__ void rhead_pack_kernel_body(const RPackArgs& a, float* __restrict__ out, int e) {
    if (e >= R_PACKED) return;
    float val = 0.f;
    if (e < RO_B1) {
        int region, local;
        if (e < RO_FW2) { region = 0; local = e; }
        else { region = 1 + (e - RO_FW2) / SZ_RHH; local = (e - RO_FW2) % SZ_RHH; }
        const int T = region == 0 ? T_RIN : T_HID;
        const int j = local & 3, ln = (local >> 2) & 63, nt = (local >> 8) & 1;
        const int t = (local >> 9) % T, w = (local >> 9) / T;
        const int n = 64 * w + 32 * nt + (ln & 31), kk = 8 * t + 4 * (ln >> 5) + j;
        switch (region) {
            case 0: val = kk < R_IN ? a.w0[n * R_IN + r_orig(kk)] : 0.f; break;
            case 1: val = a.w2[n * 256 + kk]; break;
            case 2: val = a.w2[kk * 256 + n]; break;            // g_a1[i] = sum_o G2[o] W2[o][i]
            case 3: val = a.w0[kk * R_IN + 21 + n]; break;      // g_agg[i] = sum_o G1[o] W0[o][21 + i]
            case 4: val = a.w6[n * 256 + kk]; break;            // agg[o] = sum_i W6[o][i] agg3[i]
            default: val = a.w6[kk * 256 + n]; break;           // g_agg3[i] = sum_o g_agg[o] W6[o][i]
        }
    } else if (e < RO_W3) {
        const int local = e - RO_B1;
        val = (local < 256 ? a.b0 : local < 512 ? a.b2 : a.b6)[local & 255];
    } else if (e < RO_B3) {
        val = a.w4[e - RO_W3];
    } else if (e < RO_B3 + 3) {
        val = a.b4[e - RO_B3];
    }
    out[e] = val;
}


// ---- the bf16-piece (x3) engine's fragment image ----
constexpr int RX_LDP = 296;                       // 592 B rows: 16-B aligned, 37 x 16 B (odd) -> conflict-free 16-byte reads
constexpr int RX_T1 = 18;                         // 288 / 16
constexpr int RX_TH = 16;
constexpr int RX_SZ1 = 4 * RX_T1 * 2 * 3 * 64;    // bf16x8 entries
constexpr int RX_SZH = 4 * RX_TH * 2 * 3 * 64;
constexpr int RX_FW6 = 0;
constexpr int RX_FW1 = RX_FW6 + RX_SZH;
constexpr int RX_FW2 = RX_FW1 + RX_SZ1;
constexpr int RX_BW2 = RX_FW2 + RX_SZH;
constexpr int RX_BWA = RX_BW2 + RX_SZH;
constexpr int RX_BW6 = RX_BWA + RX_SZH;
constexpr int RX_FRAGS = RX_BW6 + RX_SZH;
constexpr int RH_OFF = R_PACKED + 4 * RX_FRAGS;   // (floats) the H2 image: the same fragment order with fp16 pieces in the slots of pieces 0 and 1 (slot 2 unused)
constexpr int R_PACKED_TOTAL = R_PACKED + 8 * RX_FRAGS;

__device__ __forceinline__ void rhead_pack_x3_kernel_body(const RPackArgs& a, bf16x8* __restrict__ out, int s) {
    constexpr int N1 = RX_SZ1 / 3, NH = RX_SZH / 3;
    if (s >= N1 + 5 * NH) return;
    int region, local, base_r;      // 0 FW6, 1 FW1, 2 FW2, 3 BW2, 4 BWA, 5 BW6
    if (s < NH) { region = 0; local = s; base_r = RX_FW6; }
    else if (s < NH + N1) { region = 1; local = s - NH; base_r = RX_FW1; }
    else { region = 2 + (s - NH - N1) / NH; local = (s - NH - N1) % NH; base_r = RX_FW2 + (region - 2) * RX_SZH; }
    const int T = region == 1 ? RX_T1 : RX_TH;
    const int ln = local & 63, i = ln & 31, kg = ln >> 5, m = (local >> 6) & 1, t = (local >> 7) % T, wv = (local >> 7) / T;
    const int f = 64 * wv + 32 * m + i;
    bf16x8 p1, p2, p3, q1, q2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * t + 8 * kg + e;
        float v;
        switch (region) {
            case 0: v = a.w6[f * 256 + k]; break;                                                            // agg = W6 agg3
            case 1: v = k < 256 ? a.w0[f * R_IN + 21 + k] : (k < R_IN ? a.w0[f * R_IN + (k - 256)] : 0.f); break;
            case 2: v = a.w2[f * 256 + k]; break;
            case 3: v = a.w2[k * 256 + f]; break;                                                            // g_a1 = G2 W2
            case 4: v = a.w0[k * R_IN + 21 + f]; break;                                                      // g_agg = G1 W0[:, 21:]
            default: v = a.w6[k * 256 + f]; break;                                                           // g_agg3 = g_agg W6
        }
        __bf16 x, y, z;
        split3(v, x, y, z);
        p1[e] = x; p2[e] = y; p3[e] = z;
        const _Float16 h = (_Float16)v, g = (_Float16)((v - (float)h) * 2048.0f);
        q1[e] = __builtin_bit_cast(__bf16, h);
        q2[e] = __builtin_bit_cast(__bf16, g);
    }
    const size_t base = (size_t)base_r + (size_t)((wv * T + t) * 2 + m) * 3 * 64 + ln;
    out[base] = p1;
    out[base + 64] = p2;
    out[base + 128] = p3;
    out[RX_FRAGS + base] = q1;          // the H2 image (RH_OFF)
    out[RX_FRAGS + base + 64] = q2;
}

constexpr int R_PACK_THREADS = R_PACKED > RX_FRAGS / 3 ? R_PACKED : RX_FRAGS / 3;
__device__ __forceinline__ void rhead_pack_all(const RPackArgs& a, float* __restrict__ out, float* __restrict__ zero_buf, long long zero_floats, long long e,
                                               long long nthreads) {
    if (e < R_PACK_THREADS) {
        rhead_pack_kernel_body(a, out, (int)e);
        rhead_pack_x3_kernel_body(a, reinterpret_cast<bf16x8*>(out + R_PACKED), (int)e);
    }
    if (zero_buf) {           // the dense colour array the forward scatters into: cleared on the way (16-byte aligned: checked by the host)
        f32x4* z4 = reinterpret_cast<f32x4*>(zero_buf);
        const long long n4 = zero_floats >> 2;
        for (long long i = e; i < n4; i += nthreads) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < (zero_floats & 3)) zero_buf[(n4 << 2) + e] = 0.f;
    }
}

}  // namespace
