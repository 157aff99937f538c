// Parameter update of one optimisation step over flat fp32 arrays (K11 of DESIGN.md).
//
//  spf_adam_step   the tail of the reference's train step, spurfies/train.py:359-363 + 548-564:
//                    torch.nn.utils.clip_grad_norm_(parameters, 1.0)   -> grad *= min(1, max_norm / (|grad|_2 + 1e-6))
//                    on_after_backward()  (NaN / Inf gradient guard)   -> the update is skipped, nothing changes
//                    optimizer.step()     (torch.optim.Adam defaults)  -> exp_avg, exp_avg_sq, param
//                  as two launches (partial sums of squares; one elementwise sweep whose every block re-derives the four control numbers
//                  from the partial sums — round 4, three before) instead of ~25 foreach / reduction launches; every decision stays on the device (no host synchronisation).  (Folding the control block into the
//                  "last block to finish" of the first launch was measured SLOWER on MI355X: 21.7 us against 6.3 + 6.6 — a device-scope
//                  release / acquire per block writes back and invalidates the XCD's L2, the eight XCDs have one each.)  zero_grads: the sweep also clears the gradient buffer behind itself — the next step's
//                  optimizer.zero_grad() (train.py:357) without its fill launch.
#include <math.h>

#include "common.h"

namespace {
using namespace spf;

constexpr int NORM_BLOCKS = 512;      // partial sums: every block of the update sweep re-adds them (one wave, eight per lane; 128 blocks made the norm launch 2.3x slower)

struct AdamCtl {
    double lr, beta1, beta2, max_norm;
};

// ctl = {clip coefficient, finite flag, step size lr / (1 - beta1^t), sqrt(1 - beta2^t)};  state = {t, skipped, norm, coefficient}
// Round 4: run by the first wave of EVERY block of the update sweep (two launches instead of three: the one-wave control launch between the
// norm and the sweep cost a dispatch for ~0.5 us of work).  All blocks compute the same four numbers from the same partial sums in the same
// order; t comes from the snapshot the norm launch took (t_old), so block 0 may advance `state` while later blocks are still starting.
__device__ void adam_control(const float* __restrict__ partial, int nblk, const AdamCtl& a, const float* __restrict__ snap /* {t_old, step size, sqrt(1 - beta2^t)} */,
                             float* __restrict__ state /* block 0 only, else null */, float* __restrict__ ctl /* LDS [4] */) {
    // one wave, fixed order: lane l sums partials l, l + 64, ... in double, then a butterfly
    double s = 0.0;
    const int lane = threadIdx.x & 63;
    for (int b = lane; b < nblk; b += 64) s += (double)partial[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane != 0) return;
    const float norm = (float)sqrt(s);
    const bool finite = isfinite(norm);
    float coef = 1.f;
    if (a.max_norm > 0.0) coef = fminf((float)a.max_norm / (norm + 1e-6f), 1.0f);    // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    float t = snap[0];
    if (finite) t += 1.f;
    if (state) {
        if (!finite) state[1] += 1.f;
        state[0] = t;
        state[2] = norm;
        state[3] = coef;
    }
    ctl[0] = coef;
    ctl[1] = finite ? 1.f : 0.f;
    ctl[2] = snap[1];            // the bias corrections of step t_old + 1, formed once by the norm launch (unused when the update is skipped)
    ctl[3] = snap[2];
}


// partial[b] = sum of squares of block b's share
__global__ void __launch_bounds__(256)
sumsq_partials_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial, const float* __restrict__ state, float* __restrict__ snap, AdamCtl a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the step count BEFORE this update and the bias corrections of the step that follows it, for every block of the sweep (two double
        // pow() per BLOCK of the sweep cost it 5 us; here they cost nothing: this thread's block has 255 other threads summing)
        const float t_old = state[0];
        const double tt = (double)(t_old + 1.f);
        snap[0] = t_old;
        snap[1] = (float)(a.lr / (1.0 - pow(a.beta1, tt)));
        snap[2] = (float)sqrt(1.0 - pow(a.beta2, tt));
    }
    float s = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = g[i];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256)
adam_update_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, long long n,
                   float w1 /* 1 - beta1 */, float b2, float w2 /* 1 - beta2 */, float eps, const float* __restrict__ partial, int nblk, AdamCtl a,
                   const float* __restrict__ t_snapshot, float* __restrict__ state, int zero_grads) {
    __shared__ float ctl[4];
    if (threadIdx.x < 64) adam_control(partial, nblk, a, t_snapshot, blockIdx.x == 0 ? state : nullptr, ctl);
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (ctl[1] == 0.f) {                                          // non-finite gradient: not updating (train.py:560-564)
        if (zero_grads)
            for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) grad[i] = 0.f;
        return;
    }
    const float coef = ctl[0], step_size = ctl[2], bc2_sqrt = ctl[3];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = grad[i] * coef;
        grad[i] = zero_grads ? 0.f : g;                           // clip_grad_norm_ scales the gradients in place; or: cleared for the next step
        float mi = m[i], vi = v[i];
        mi = mi + w1 * (g - mi);                                  // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * b2;                                             // exp_avg_sq.mul_(beta2)
        vi = vi + (w2 * g) * g;                                   //           .addcmul_(grad, grad, value=1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[i] = param[i] + (-step_size) * (mi / denom);        // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}

}  // namespace

extern "C" {

int64_t spf_adam_workspace_floats(void) { return NORM_BLOCKS + 4; }      // partial sums, then {t_old, step size, sqrt(1 - beta2^t)}

int spf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                  double max_norm, int32_t zero_grads, float* state, float* workspace, void* stream) {
    if (n < 0 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return spf::fail(SPF_EINVAL, "spf_adam_step: need n >= 0, 0 <= beta < 1, eps >= 0");
    if (n == 0) return SPF_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || !workspace) return spf::fail(SPF_EINVAL, "spf_adam_step: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int nblk = spf::div_up(n, 256 * 8);
    nblk = nblk < 1 ? 1 : (nblk > NORM_BLOCKS ? NORM_BLOCKS : nblk);
    float* t_snapshot = workspace + NORM_BLOCKS;
    sumsq_partials_kernel<<<nblk, 256, 0, s>>>(grad, n, workspace, state, t_snapshot, AdamCtl{lr, beta1, beta2, max_norm});
    int ublk = spf::div_up(n, 256 * 4);
    ublk = ublk > 2048 ? 2048 : ublk;
    adam_update_kernel<<<ublk, 256, 0, s>>>(param, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, workspace,
                                            nblk, AdamCtl{lr, beta1, beta2, max_norm}, t_snapshot, state, zero_grads ? 1 : 0);
    SPF_LAUNCH_CHECK("spf_adam_step");
    return SPF_OK;
}

}  // extern "C"
