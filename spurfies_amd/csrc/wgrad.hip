// Weight-gradient GEMM with a DEVICE-side row count:  dW[256, C] += G[rows, 256]^T * A[rows, C],  rows = *n_rows.
//
// The reference gets these from autograd's AddmmBackward (one cuBLAS GEMM per nn.Linear of F_color / R,
// spurfies/model/pointneus_disent.py:76-107).  Here the K dimension (rows = valid pairs or points) is only known on
// the device, so a library GEMM would force a host round trip per step; this kernel reads the count itself, which
// keeps the whole optimisation step free of host synchronisation (and capturable in a hipGraph).
//
// K-huge / MN-tiny shape: every workgroup owns a slice of the rows and the FULL 256 x C output in accumulators
// (one wave per SIMD, 2 x NT tiles of v_mfma_f32_32x32x2_f32 = up to 256 VGPRs), streams G and A straight from HBM
// (each element read once; per row pair a wave issues one float2 + NT/4 float4 loads for 2*NT MFMAs) and leaves its
// partial in a slab; a second tiny kernel sums the slabs in a fixed order (bitwise reproducible) into dW.
#include "mlp_tile.h"

namespace {
using namespace spf;

// Operand maps (chosen so that every load is a wide, fully coalesced one):
//   A operand (G^T): tile m, lane (o_l = lane & 31)  <->  output row o = 64 wave + 2 o_l + m   (one float2 load per row)
//   B operand (A):   tile t, lane (c_i = lane & 31)  <->  column     i = NT c_i + t            (NT/4 float4 loads per row)
// Each workgroup writes its partial [256][32 NT] in that native accumulator order to a slab; wgrad_reduce_kernel
// sums the slabs of the active workgroups in a fixed order (deterministic) and adds the result to dW.
template <int NT>
__global__ void __launch_bounds__(256, 1)
wgrad_kernel(const float* __restrict__ G, const float* __restrict__ A, int lda, int C, const int32_t* __restrict__ n_rows_dev,
             int max_rows, float* __restrict__ slab) {
    const int lane = threadIdx.x & 63, ci = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;   // even-sized row slices: a k-pair never straddles two workgroups
    chunk += chunk & 1;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, n);
    if (r0 >= r1) return;
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    const float* gp = G + 64 * wave + 2 * ci;
    const bool bok = NT * ci < C;            // C is a multiple of NT for every caller (256 / 104 / 21 with NT 8 / 4 / 1)
    constexpr int U = (NT == 8) ? 6 : 8;     // row pairs per software-pipeline stage
    float2 a[U], an[U];
    float b[U][NT], bn[U][NT];
    auto load = [&](int base, float2 (&fa)[U], float (&fb)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = base + 2 * u + h;
            const bool ok = row < r1;
            const size_t rr = ok ? (size_t)row : (size_t)r0;
            fa[u] = *reinterpret_cast<const float2*>(gp + rr * 256);
            if (!ok) fa[u] = make_float2(0.f, 0.f);
            const float* ap = A + rr * lda + NT * ci;
            if (NT >= 4) {
#pragma unroll
                for (int v = 0; v < NT / 4; ++v) {
                    f32x4 x4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok && bok) x4 = *reinterpret_cast<const f32x4*>(ap + 4 * v);
                    fb[u][4 * v] = x4[0]; fb[u][4 * v + 1] = x4[1]; fb[u][4 * v + 2] = x4[2]; fb[u][4 * v + 3] = x4[3];
                }
            } else {
                fb[u][0] = (ok && bok) ? ap[0] : 0.f;
            }
        }
    };
    load(r0, a, b);
    for (int base = r0; base < r1; base += 2 * U) {
        load(base + 2 * U, an, bn);          // next stage's operands are in flight while this stage's MFMAs issue
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u][t], acc[1][t], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = an[u];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[u][t] = bn[u][t];
        }
    }
    // slab[block][wave][m][t][reg][lane]
    float* out = slab + ((size_t)blockIdx.x * 4 + wave) * (2 * NT * 16 * 64) + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((m * NT + t) * 16 + r) * 64] = acc[m][t][r];
}

template <int NT>
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nblk_launched, const int32_t* __restrict__ n_rows_dev, int max_rows, int C,
                                    float* __restrict__ dW, int ldw) {
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    if (n <= 0) return;
    int chunk = (n + nblk_launched - 1) / nblk_launched;
    chunk += chunk & 1;
    const int active = (n + chunk - 1) / chunk;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // element of one slab: [wave][m][t][reg][lane]
    constexpr int PER = 4 * 2 * NT * 16 * 64;
    if (e >= PER) return;
    const int lane = e & 63, r = (e >> 6) & 15, mt = (e >> 10) % (2 * NT), wave = (e >> 10) / (2 * NT);
    const int m = mt / NT, t = mt % NT;
    const int o = 64 * wave + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + m;
    const int i = NT * (lane & 31) + t;
    if (i >= C) return;
    float s = 0.f;
    for (int b = 0; b < active; ++b) s += slab[(size_t)b * PER + e];
    dW[(size_t)o * ldw + i] += s;
}

}  // namespace

extern "C" {

int64_t spf_wgrad_workspace_floats(int32_t C) { return (int64_t)256 * 256 * (C > 128 ? 256 : (C > 32 ? 128 : 32)); }

int spf_wgrad(const float* G, const float* A, int32_t lda, int32_t C, const int32_t* n_rows, int32_t max_rows, float* dW, int32_t ldw,
              float* workspace, void* stream) {
    if (max_rows < 0 || C < 1 || C > 256 || lda < C || ldw < C) return spf::fail(SPF_EINVAL, "spf_wgrad: need 1 <= C <= 256, lda >= C, ldw >= C");
    if (max_rows == 0) return SPF_OK;
    if (!G || !A || !dW || !workspace) return spf::fail(SPF_EINVAL, "spf_wgrad: null pointer");
    const int NT = C > 128 ? 8 : (C > 32 ? 4 : 1);
    if (C % NT || (NT >= 4 && (lda % 4))) return spf::fail(SPF_EINVAL, "spf_wgrad: C must be a multiple of %d and lda of 4 (C=%d lda=%d)", NT, C, lda);
    hipStream_t s = (hipStream_t)stream;
    int blocks = spf::div_up(max_rows, 512);
    if (blocks > 256) blocks = 256;   // one workgroup per CU, one wave per SIMD
    const int per = 4 * 2 * NT * 16 * 64;
    if (NT == 8) {
        wgrad_kernel<8><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<8><<<spf::div_up(per, 256), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    } else if (NT == 4) {
        wgrad_kernel<4><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<4><<<spf::div_up(per, 256), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    } else {
        wgrad_kernel<1><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<1><<<spf::div_up(per, 256), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    }
    SPF_LAUNCH_CHECK("wgrad_kernel");
    return SPF_OK;
}

}  // extern "C"
