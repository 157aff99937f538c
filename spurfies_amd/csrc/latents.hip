// Latent-table kernels (K8 of DESIGN.md): row scatter-add and the total-variation regulariser.
//
//  spf_scatter_add_rows   backward of the reference's index_select gathers
//                         (spurfies/model/utils.py:158-161) — replaces PyTorch's sort-based
//                         index_put_(accumulate) kernel, which dominated the first profile.
//  spf_tv_forward/backward  spurfies/model/utils.py:221-282 on the static neighbour graph.
#include "common.h"

namespace {
using namespace spf;

// dst[idx[m], :] += src[m, :]  — C/4 lanes per row (float4), float atomics
template <int C>
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, long long M,
                                        float* __restrict__ dst) {
    constexpr int LPR = C / 4;  // lanes per row
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = gid / LPR;
    const int c = (int)(gid % LPR) * 4;
    if (m >= M) return;
    const int i = idx[m];
    if (i < 0) return;
    const float4 v = *reinterpret_cast<const float4*>(src + m * C + c);
    float* d = dst + (size_t)i * C + c;
    atomicAdd(d, v.x);
    atomicAdd(d + 1, v.y);
    atomicAdd(d + 2, v.z);
    atomicAdd(d + 3, v.w);
}

// one 32-lane group per point; lane = latent channel.  tv_i = sum_j w_ij |f_j - f_i|_1 / norm_i
__global__ void tv_forward_kernel(const float* __restrict__ feat, const int32_t* __restrict__ nbr, const float* __restrict__ w,
                                  const float* __restrict__ norm, int n, int k, float* __restrict__ tv) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 5, c = gid & 31;
    if (i >= n) return;
    const float fi = feat[(size_t)i * 32 + c];
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const float wj = w[(size_t)i * k + j];
        if (wj == 0.f) continue;
        const int q = nbr[(size_t)i * k + j];
        acc += wj * fabsf(feat[(size_t)q * 32 + c] - fi);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (c == 0) tv[i] = acc / norm[i];
}

// g_feat[i] += s_i * sign(f_i - f_j) * w_ij ; g_feat[j] -= same,  s_i = g_tv[i] / norm_i
__global__ void tv_backward_kernel(const float* __restrict__ feat, const int32_t* __restrict__ nbr, const float* __restrict__ w,
                                   const float* __restrict__ norm, const float* __restrict__ g_tv, int g_tv_stride, float scale, int n, int k,
                                   float* __restrict__ g_feat, long long* __restrict__ g_fixed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 5, c = gid & 31;
    if (i >= n) return;
    const float fi = feat[(size_t)i * 32 + c];
    const float s = g_tv[(size_t)i * g_tv_stride] * scale / norm[i];
    float own = 0.f;
    for (int j = 0; j < k; ++j) {
        const float wj = w[(size_t)i * k + j];
        if (wj == 0.f) continue;
        const int q = nbr[(size_t)i * k + j];
        const float d = feat[(size_t)q * 32 + c] - fi;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // d|d|/dd, 0 at 0 like torch.abs
        const float g = s * wj * sg;
        if (g_fixed) fixed_add(g_fixed, (size_t)q * 32 + c, g);
        else atomicAdd(&g_feat[(size_t)q * 32 + c], g);
        own -= g;
    }
    if (g_fixed) fixed_add(g_fixed, (size_t)i * 32 + c, own);
    else atomicAdd(&g_feat[(size_t)i * 32 + c], own);
}

// dst[i] += acc[i] * 2^-48 (the exact sum, rounded to fp32 once); acc[i] = 0 for the next use.  A set status word (bit 0 of acc[-1]:
// some term was non-finite, common.h) turns every dst[i] into NaN.  The word is cleared by this launch itself: every workgroup reads it
// first, takes a ticket in its upper half when it is done, and the LAST workgroup to finish stores zero — no memset behind the kernel (a
// hipMemsetAsync here became a memset node of a captured step; on ROCm 7.2 such a graph, replayed after a hipDeviceSynchronize, flushed NaN
// on every later step — found by tools/soak.py's resume leg).
__global__ void fixed_accumulate_kernel(long long* __restrict__ acc, float* __restrict__ dst, long long n) {
    __shared__ int s_poisoned;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(acc - 1);
    if (threadIdx.x == 0) s_poisoned = (int)(__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1ull);
    __syncthreads();
    const bool poisoned = s_poisoned != 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long a = acc[i];
        if (poisoned) {
            dst[i] = __builtin_nanf("");
            acc[i] = 0;
        } else if (a != 0) {
            const double d = (double)a;
            dst[i] += (fabs(d) >= 0.5 * FIXED_LIMIT) ? __builtin_nanf("") : (float)(d / FIXED_SCALE);
            acc[i] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(status, 1ull << 32) >> 32;       // every workgroup has read the flag before it takes a ticket
        if (t == (unsigned long long)gridDim.x - 1) __hip_atomic_store(status, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

extern "C" {

int spf_scatter_add_rows(const float* src, const int32_t* idx, int64_t m, int32_t c, float* dst, void* stream) {
    if (m < 0 || (c != 32 && c != 64 && c != 4)) return spf::fail(SPF_EINVAL, "spf_scatter_add_rows: c must be 4, 32 or 64 (got %d)", c);
    if (m == 0) return SPF_OK;
    if (!src || !idx || !dst) return spf::fail(SPF_EINVAL, "spf_scatter_add_rows: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long threads = (long long)m * (c / 4);
    const int blocks = spf::div_up(threads, 256);
    if (c == 64) scatter_add_rows_kernel<64><<<blocks, 256, 0, s>>>(src, idx, m, dst);
    else if (c == 32) scatter_add_rows_kernel<32><<<blocks, 256, 0, s>>>(src, idx, m, dst);
    else scatter_add_rows_kernel<4><<<blocks, 256, 0, s>>>(src, idx, m, dst);
    SPF_LAUNCH_CHECK("scatter_add_rows_kernel");
    return SPF_OK;
}

int spf_tv_forward(const float* feat_geo, const int32_t* nbr, const float* w, const float* norm, int32_t n, int32_t k,
                   float* tv, void* stream) {
    if (n < 0 || k < 1) return spf::fail(SPF_EINVAL, "spf_tv_forward: bad sizes");
    if (n == 0) return SPF_OK;
    if (!feat_geo || !nbr || !w || !norm || !tv) return spf::fail(SPF_EINVAL, "spf_tv_forward: null pointer");
    tv_forward_kernel<<<spf::div_up((long long)n * 32, 256), 256, 0, (hipStream_t)stream>>>(feat_geo, nbr, w, norm, n, k, tv);
    SPF_LAUNCH_CHECK("tv_forward_kernel");
    return SPF_OK;
}

int spf_tv_backward(const float* feat_geo, const int32_t* nbr, const float* w, const float* norm, const float* g_tv,
                    int32_t g_tv_stride, float scale, int32_t n, int32_t k, float* g_feat_geo, int64_t* g_feat_geo_fixed, void* stream) {
    if (n < 0 || k < 1) return spf::fail(SPF_EINVAL, "spf_tv_backward: bad sizes");
    if (n == 0) return SPF_OK;
    if (!feat_geo || !nbr || !w || !norm || !g_tv || (!g_feat_geo && !g_feat_geo_fixed)) return spf::fail(SPF_EINVAL, "spf_tv_backward: null pointer");
    if (g_tv_stride != 0 && g_tv_stride != 1) return spf::fail(SPF_EINVAL, "spf_tv_backward: g_tv_stride is 1 (per point) or 0 (one value for all points)");
    tv_backward_kernel<<<spf::div_up((long long)n * 32, 256), 256, 0, (hipStream_t)stream>>>(feat_geo, nbr, w, norm, g_tv, g_tv_stride, scale, n, k, g_feat_geo,
                                                                                           reinterpret_cast<long long*>(g_feat_geo_fixed));
    SPF_LAUNCH_CHECK("tv_backward_kernel");
    return SPF_OK;
}

int spf_fixed_accumulate(int64_t* acc, float* dst, int64_t n, void* stream) {
    if (n < 0) return spf::fail(SPF_EINVAL, "spf_fixed_accumulate: bad size");
    if (n == 0) return SPF_OK;
    if (!acc || !dst) return spf::fail(SPF_EINVAL, "spf_fixed_accumulate: null pointer");
    int blocks = spf::div_up(n, 256);
    if (blocks > 4096) blocks = 4096;
    fixed_accumulate_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<long long*>(acc), dst, (long long)n);
    SPF_LAUNCH_CHECK("fixed_accumulate_kernel");
    return SPF_OK;
}

}  // extern "C"
