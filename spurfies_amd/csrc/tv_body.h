// Total-variation regulariser over the static neighbour graph (spurfies/model/utils.py:221-282) as device functions, so that the forward can
// ride in the step's first launch (camera.hip: spf_step_prologue) and the backward in the loss backward launch (loss.hip) beside
// latents.hip's own kernels: one 32-lane group per point, lane = latent channel.
#pragma once
#include "common.h"

namespace spf {

struct TvArgs {
    const float* feat;      // [n,32] geometry latents; NULL = no TV work
    const int32_t* nbr;     // [n,k]
    const float* w;         // [n,k] inverse-distance weights (0 = absent)
    const float* norm;      // [n]
    int n, k;
    float* tv;              // forward: [n] per-point terms
    float* g_feat;          // backward: [n,32] accumulated with float atomics
};

#ifdef __HIPCC__
// tv_i = sum_j w_ij |f_j - f_i|_1 / norm_i          (gid = 32 i + channel)
__device__ __forceinline__ void tv_forward_body(const TvArgs& a, long long gid) {
    const int i = (int)(gid >> 5), c = (int)(gid & 31);
    if (i >= a.n) return;
    const float fi = a.feat[(size_t)i * 32 + c];
    float acc = 0.f;
    for (int j = 0; j < a.k; ++j) {
        const float wj = a.w[(size_t)i * a.k + j];
        if (wj == 0.f) continue;
        const int q = a.nbr[(size_t)i * a.k + j];
        acc += wj * fabsf(a.feat[(size_t)q * 32 + c] - fi);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (c == 0) a.tv[i] = acc / a.norm[i];
}

// g_feat[i] += s_i * sign(f_i - f_j) * w_ij ; g_feat[j] -= same,  s_i = g / norm_i  (g = d loss / d tv_i, the same for every point here)
__device__ __forceinline__ void tv_backward_body(const TvArgs& a, long long gid, float g) {
    const int i = (int)(gid >> 5), c = (int)(gid & 31);
    if (i >= a.n) return;
    const float fi = a.feat[(size_t)i * 32 + c];
    const float s = g / a.norm[i];
    float own = 0.f;
    for (int j = 0; j < a.k; ++j) {
        const float wj = a.w[(size_t)i * a.k + j];
        if (wj == 0.f) continue;
        const int q = a.nbr[(size_t)i * a.k + j];
        const float d = a.feat[(size_t)q * 32 + c] - fi;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // d|d|/dd, 0 at 0 like torch.abs
        const float gg = s * wj * sg;
        atomicAdd(&a.g_feat[(size_t)q * 32 + c], gg);
        own -= gg;
    }
    atomicAdd(&a.g_feat[(size_t)i * 32 + c], own);
}
#endif

}  // namespace spf
