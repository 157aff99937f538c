// VolSDF loss terms of one optimisation step, fused (K10 of DESIGN.md).
//
//  spf_loss_forward    spurfies/model/loss.py:42-49,51-101 (rgb L1 mean, eikonal mean, mask BCE on clip(sum w), weighted total) and
//                      the pseudo-point term of pointneus_disent.py:765-780, from the renderer's dense per-ray outputs:
//                      one partial-sum launch over rays / shading slots + one single-block launch that forms the terms.
//  spf_loss_backward   their gradients w.r.t. rgb [R,3], acc = sum_j w_j [R], the pseudo-point SDF [R] and the TV term.
// For ray-sharded steps the caller passes the all-reduced (global) counts in `denom`, so that the ranks' losses sum to the
// single-GPU batch loss (spurfies_amd/dist.py).  ~70 elementwise / reduction launches of the PyTorch formulation -> 3.
#include "common.h"
#include "tv_body.h"

namespace {
using namespace spf;

constexpr int NPART = 8;       // floats per partial row
constexpr int MAX_BLOCKS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// the feature-consistency term's per-ray values (spf_local_forward), by value in the kernel arguments
struct LocalIn {
    const float* lsum;
    const int32_t* lfirst;
    const spf_local_desc* desc;
};
__host__ inline LocalIn local_in(const spf_local_terms* l) { return l ? LocalIn{l->lsum, l->lfirst, l->desc} : LocalIn{nullptr, nullptr, nullptr}; }

// partial[b] = {sum |rgb - gt|, sum BCE, sum_valid (|g| - 1)^2, sum_use |psdf|, sum use, sum tv_i, sum lsum, n_src * #crossings} over block b's
// grid-stride share
__global__ void __launch_bounds__(256)
loss_partials_kernel(const float* __restrict__ rgb, const float* __restrict__ rgb_gt, const float* __restrict__ acc,
                     const float* __restrict__ mask_gt, int mstride, const float* __restrict__ grad, const uint8_t* __restrict__ slot_valid,
                     const float* __restrict__ psdf, const uint8_t* __restrict__ pvalid, const uint8_t* __restrict__ ray_valid, int R,
                     long long rows, const float* __restrict__ tv, int n_tv, LocalIn loc, float* __restrict__ partial) {
    float s[NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float n_src = loc.lsum ? (float)loc.desc->n_src : 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long r = t0; r < R; r += stride) {
        s[0] += fabsf(rgb[3 * r] - rgb_gt[3 * r]) + fabsf(rgb[3 * r + 1] - rgb_gt[3 * r + 1]) + fabsf(rgb[3 * r + 2] - rgb_gt[3 * r + 2]);
        const float a = fminf(fmaxf(acc[r], 1e-3f), 1.0f - 1e-3f), m = mask_gt[(size_t)r * mstride];
        s[1] -= m * fmaxf(logf(a), -100.f) + (1.f - m) * fmaxf(logf(1.f - a), -100.f);   // F.binary_cross_entropy's clamped logs
        if (psdf && pvalid[r] && ray_valid[r]) {
            s[3] += fabsf(psdf[r]);
            s[4] += 1.f;
        }
        if (loc.lsum && loc.lfirst[r] >= 0) {        // feat_utils.py:437: the mean runs over source views x rays with a crossing
            s[6] += loc.lsum[r];
            s[7] += n_src;
        }
    }
    if (grad)
        for (long long q = t0; q < rows; q += stride)
            if (slot_valid[q]) {
                const float gx = grad[3 * q], gy = grad[3 * q + 1], gz = grad[3 * q + 2];
                const float d = sqrtf(gx * gx + gy * gy + gz * gz) - 1.f;
                s[2] += d * d;
            }
    for (long long i = t0; i < n_tv; i += stride) s[5] += tv[i];            // the per-point TV terms (spf_tv_forward): their mean is formed here
    __shared__ float red[4][NPART];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NPART; ++i) {
        const float v = wave_sum(s[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NPART)
        partial[blockIdx.x * NPART + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out = {loss, rgb, eikonal, tv, mask, local, pseudo, pseudo count};  den = {1/(3 R), 1/R, 1/pseudo count or 0, 1/world, 1/max(local count, 1)}
// (the first wave of the calling block; lane 0 writes.  den_out: 4 floats anywhere — global memory, or LDS for the fused backward)
__device__ __forceinline__ void loss_finalize_wave(const float* __restrict__ partial, int nblk, int R, const int32_t* __restrict__ n_points,
                                                   const float* __restrict__ tv, int n_tv, const float* __restrict__ denom, const spf_loss_weights& w,
                                                   float* __restrict__ total, float* __restrict__ out, float* den, float* den2, bool has_local) {
    const int lane = threadIdx.x & 63;
    float s[NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < nblk; b += 64)
#pragma unroll
        for (int i = 0; i < NPART; ++i) s[i] += partial[b * NPART + i];
#pragma unroll
    for (int i = 0; i < NPART; ++i) s[i] = wave_sum(s[i]);
    if (lane != 0) return;
    const float G = (float)(w.world > 0 ? w.world : 1);
    const float R_tot = denom ? denom[0] : (float)R;
    const float P_tot = fmaxf(denom ? denom[1] : (n_points ? (float)*n_points : 0.f), 1.f);
    const float ps_tot = denom ? denom[2] : s[4];
    const float l_rgb = s[0] / (3.0f * R_tot);
    const float l_mask = s[1] / R_tot;
    const float l_eik = s[2] / P_tot;
    const float l_tv = (tv && w.tv > 0.f) ? (n_tv > 0 ? s[5] / (float)n_tv : *tv) / G : 0.f;      // n_tv > 0: tv is the per-point array, else the mean itself
    // no rendered point with a neighbour on any rank: the reference's constant 1000 (pointneus_disent.py:776-780)
    const float l_pseudo = w.pseudo > 0.f ? (ps_tot > 0.f ? s[3] / ps_tot : 1000.0f / G) : 0.f;
    // feature-consistency term: sum over this rank's rays / (source views x crossings of the whole batch); 0 without a crossing
    // (feat_utils.py:390-391)
    const float lc_tot = fmaxf(has_local ? (denom ? denom[3] : s[7]) : 0.f, 1.f);
    const float l_local = (has_local && w.local > 0.f) ? s[6] / lc_tot : 0.f;
    const float loss = w.rgb * l_rgb + w.eikonal * l_eik + w.tv * l_tv + w.local * l_local + w.pseudo * l_pseudo + l_mask;
    if (total) {
        *total = loss;
        out[0] = loss; out[1] = l_rgb; out[2] = l_eik; out[3] = l_tv; out[4] = l_mask; out[5] = l_local; out[6] = l_pseudo; out[7] = s[4];
    }
    const float d0 = 1.0f / (3.0f * R_tot), d1 = 1.0f / R_tot, d2 = ps_tot > 0.f ? 1.0f / ps_tot : 0.f, d3 = 1.0f / G, d4 = 1.0f / lc_tot;
    if (den) { den[0] = d0; den[1] = d1; den[2] = d2; den[3] = d3; den[4] = d4; }
    if (den2) { den2[0] = d0; den2[1] = d1; den2[2] = d2; den2[3] = d3; den2[4] = d4; }
}

__global__ void __launch_bounds__(64)
loss_finalize_kernel(const float* __restrict__ partial, int nblk, int R, const int32_t* __restrict__ n_points, const float* __restrict__ tv, int n_tv,
                     const float* __restrict__ denom, spf_loss_weights w, float* __restrict__ total, float* __restrict__ out,
                     float* __restrict__ den, bool has_local) {
    loss_finalize_wave(partial, nblk, R, n_points, tv, n_tv, denom, w, total, out, den, nullptr, has_local);
}

// fin.partial != NULL: the forward ran its partial-sum launch only (spf_loss_forward with total == NULL); every block of this launch then forms
// the normalisers itself from the <= 256 partial rows (first wave, into LDS) and block 0 also writes the loss terms — the single-block
// finalize launch between the partial sums and the backward is gone (round 5: 3 launches -> 2 per step).
struct LossFin {
    const float* partial;
    int nblk, R;
    const int32_t* n_points;
    const float* tv;
    int n_tv;
    const float* denom;
    float *total, *terms, *den_out;
    bool has_local;
};
__global__ void loss_backward_kernel(const float* __restrict__ g_total, const float* __restrict__ den_in, spf_loss_weights w,
                                     const float* __restrict__ rgb, const float* __restrict__ rgb_gt, const float* __restrict__ acc,
                                     const float* __restrict__ mask_gt, int mstride, const float* __restrict__ psdf, const uint8_t* __restrict__ pvalid,
                                     const uint8_t* __restrict__ ray_valid, int R, float* __restrict__ g_rgb, float* __restrict__ g_acc,
                                     float* __restrict__ g_psdf, float* __restrict__ g_tv, int n_tv, LossFin fin, int ray_blocks, TvArgs tvb,
                                     float* __restrict__ lscale) {
    __shared__ float s_den[8];
    const float* den = den_in;
    if (fin.partial) {
        if (threadIdx.x < 64)
            loss_finalize_wave(fin.partial, fin.nblk, fin.R, fin.n_points, fin.tv, fin.n_tv, fin.denom, w, blockIdx.x == 0 ? fin.total : nullptr, fin.terms,
                               blockIdx.x == 0 ? fin.den_out : nullptr, s_den, fin.has_local);
        __syncthreads();
        den = s_den;
    }
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const float g = *g_total;
    // d loss / d (TV mean), or — n_tv > 0 — d loss / d tv_i, the same for every point
    const float gtv = w.tv > 0.f ? g * w.tv * den[3] / (n_tv > 0 ? (float)n_tv : 1.f) : 0.f;
    if ((int)blockIdx.x >= ray_blocks) {       // the TV term's backward rides behind the ray blocks (tv_body.h): one launch less per step
        if (gtv != 0.f) tv_backward_body(tvb, (long long)((int)blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x, gtv);
        return;
    }
    if (r == 0 && g_tv) *g_tv = gtv;
    if (r == 0 && lscale) *lscale = g * w.local * den[4];      // what spf_render_backward multiplies the crossings' coefficients by
    if (r >= R) return;
    const float c_rgb = g * w.rgb * den[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = rgb[3 * r + c] - rgb_gt[3 * r + c];
        g_rgb[3 * r + c] = c_rgb * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    const float a0 = acc[r], m = mask_gt[(size_t)r * mstride];
    float ga = 0.f;
    if (a0 >= 1e-3f && a0 <= 1.0f - 1e-3f)                                           // clamp passes the gradient inside [min, max]
        ga = g * den[1] * (a0 - m) / fmaxf((1.f - a0) * a0, 1e-12f);                 // binary_cross_entropy_backward
    g_acc[r] = ga;
    if (g_psdf) {
        float gp = 0.f;
        if (psdf && pvalid[r] && ray_valid[r]) {
            const float v = psdf[r];
            gp = g * w.pseudo * den[2] * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
        }
        g_psdf[r] = gp;
    }
}

}  // namespace

extern "C" {

int64_t spf_loss_workspace_floats(void) { return (int64_t)MAX_BLOCKS * NPART; }

// workgroups of the partial-sum launch (both entry points derive it from the same sizes)
static int loss_blocks(int R, long long rows, int n_tv) {
    long long work = rows > R ? rows : R;
    if (n_tv > work) work = n_tv;
    int nblk = spf::div_up(work, 256 * 4);
    return nblk < 1 ? 1 : (nblk > MAX_BLOCKS ? MAX_BLOCKS : nblk);
}

int spf_loss_forward(const float* rgb, const float* rgb_gt, const float* acc, const float* mask_gt, int32_t mask_stride, const float* grad,
                     const uint8_t* slot_valid, int64_t rows, const int32_t* n_points, const float* psdf, const uint8_t* pvalid,
                     const uint8_t* ray_valid, const float* tv, int32_t n_tv, const float* denom, int32_t R, const spf_loss_weights* weights,
                     float* workspace, float* total, float* terms, float* den, const spf_local_terms* local, void* stream) {
    if (R <= 0 || rows < 0 || !weights || mask_stride < 1 || n_tv < 0) return spf::fail(SPF_EINVAL, "spf_loss_forward: need R > 0, rows >= 0, mask_stride >= 1, n_tv >= 0, weights");
    if (!rgb || !rgb_gt || !acc || !mask_gt || !workspace) return spf::fail(SPF_EINVAL, "spf_loss_forward: null pointer");
    if ((total == nullptr) != (terms == nullptr) || (total == nullptr) != (den == nullptr))
        return spf::fail(SPF_EINVAL, "spf_loss_forward: total, terms and den are given (or all left out: partial sums only) together");
    if (grad && !slot_valid) return spf::fail(SPF_EINVAL, "spf_loss_forward: grad needs slot_valid");
    if (psdf && (!pvalid || !ray_valid)) return spf::fail(SPF_EINVAL, "spf_loss_forward: psdf needs pvalid and ray_valid");
    if (local && (!local->lsum || !local->lfirst || !local->desc)) return spf::fail(SPF_EINVAL, "spf_loss_forward: local needs lsum, lfirst and desc");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = loss_blocks(R, grad ? rows : 0, tv ? n_tv : 0);
    loss_partials_kernel<<<nblk, 256, 0, s>>>(rgb, rgb_gt, acc, mask_gt, mask_stride, grad, slot_valid, psdf, pvalid, ray_valid, R, rows, tv, tv ? n_tv : 0,
                                           local_in(local), workspace);
    if (total) loss_finalize_kernel<<<1, 64, 0, s>>>(workspace, nblk, R, n_points, tv, n_tv, denom, *weights, total, terms, den, local != nullptr);
    SPF_LAUNCH_CHECK("loss_forward");
    return SPF_OK;
}

int spf_loss_backward(const float* g_total, const float* den, const spf_loss_weights* weights, const float* rgb, const float* rgb_gt,
                      const float* acc, const float* mask_gt, int32_t mask_stride, const float* psdf, const uint8_t* pvalid, const uint8_t* ray_valid,
                      int32_t R, float* g_rgb, float* g_acc, float* g_psdf, float* g_tv, int32_t n_tv, const spf_local_terms* local, void* stream) {
    if (R <= 0 || !weights) return spf::fail(SPF_EINVAL, "spf_loss_backward: need R > 0, weights");
    if (!g_total || !den || !rgb || !rgb_gt || !acc || !mask_gt || !g_rgb || !g_acc) return spf::fail(SPF_EINVAL, "spf_loss_backward: null pointer");
    loss_backward_kernel<<<spf::div_up(R, 256), 256, 0, (hipStream_t)stream>>>(g_total, den, *weights, rgb, rgb_gt, acc, mask_gt, mask_stride, psdf, pvalid,
                                                                               ray_valid, R, g_rgb, g_acc, g_psdf, g_tv, n_tv, LossFin{}, spf::div_up(R, 256), TvArgs{},
                                                                               local ? local->lscale : nullptr);
    SPF_LAUNCH_CHECK("loss_backward_kernel");
    return SPF_OK;
}

int spf_loss_backward_finalize(const float* g_total, const spf_loss_weights* weights, const float* rgb, const float* rgb_gt, const float* acc,
                               const float* mask_gt, int32_t mask_stride, const float* psdf, const uint8_t* pvalid, const uint8_t* ray_valid, int32_t R,
                               float* g_rgb, float* g_acc, float* g_psdf, float* g_tv, int32_t n_tv, const float* workspace, int64_t rows,
                               const int32_t* n_points, const float* tv, const float* denom, float* total, float* terms, float* den,
                               const float* tv_feat, const int32_t* tv_nbr, const float* tv_w, const float* tv_norm, int32_t tv_k, float* tv_g_feat,
                               const spf_local_terms* local, void* stream) {
    if (R <= 0 || !weights || rows < 0 || n_tv < 0) return spf::fail(SPF_EINVAL, "spf_loss_backward_finalize: need R > 0, rows >= 0, n_tv >= 0, weights");
    if (!g_total || !rgb || !rgb_gt || !acc || !mask_gt || !g_rgb || !g_acc || !workspace || !total || !terms || !den)
        return spf::fail(SPF_EINVAL, "spf_loss_backward_finalize: null pointer");
    LossFin fin{workspace, loss_blocks(R, rows, tv ? n_tv : 0), R, n_points, tv, n_tv, denom, total, terms, den, local != nullptr};
    TvArgs tvb{};
    int tv_blocks = 0;
    if (tv_g_feat) {          // the TV term's backward in the same launch: g_feat += d loss / d tv_i * d tv_i / d latents
        if (!tv_feat || !tv_nbr || !tv_w || !tv_norm || tv_k < 1 || n_tv < 1 || !tv) return spf::fail(SPF_EINVAL, "spf_loss_backward_finalize: incomplete TV arguments (need the per-point form, n_tv > 0)");
        tvb = TvArgs{tv_feat, tv_nbr, tv_w, tv_norm, n_tv, tv_k, nullptr, tv_g_feat};
        tv_blocks = spf::div_up((long long)n_tv * 32, 256);
    }
    const int ray_blocks = spf::div_up(R, 256);
    loss_backward_kernel<<<ray_blocks + tv_blocks, 256, 0, (hipStream_t)stream>>>(g_total, nullptr, *weights, rgb, rgb_gt, acc, mask_gt, mask_stride, psdf, pvalid,
                                                                                  ray_valid, R, g_rgb, g_acc, g_psdf, g_tv, n_tv, fin, ray_blocks, tvb,
                                                                                  local ? local->lscale : nullptr);
    SPF_LAUNCH_CHECK("loss_backward_kernel<finalize>");
    return SPF_OK;
}

}  // extern "C"
