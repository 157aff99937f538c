// Per-ray compositing (K6 of DESIGN.md): slot depths + deltas, Laplace density, alpha compositing and
// the rendered quantities, forward and backward.  One wavefront per ray; the exclusive cumulative
// sum over the <= 80 slots is a wave prefix-sum (two 64-lane chunks with a carry).
//
// Replaces
//   filter_points     spurfies/model/pointneus_disent.py:207-239
//   LaplaceDensity    spurfies/model/density.py:16-30
//   volume_rendering  spurfies/model/pointneus_disent.py:894-908
//   composites        spurfies/model/pointneus_disent.py:765-795 (dist_map, rgb, depth, acc)
// and autograd's backward through them (the reference runs ~60 elementwise / cumsum launches).
#include "common.h"
#include "grid_dev.h"

namespace {
using namespace spf;

// ---- filter_points: one thread per slot (grid_dev.h: filter_slot) -----------------------------------------------------------------
__global__ void filter_points_kernel(const float* __restrict__ loc, const uint8_t* __restrict__ valid,
                                     const float* __restrict__ cam_loc, const float* __restrict__ ray_dirs, int R, int SR,
                                     float* __restrict__ z, float* __restrict__ deltas, float* __restrict__ x) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)R * SR) return;
    filter_slot(gid, valid, FilterArgs{loc, cam_loc, ray_dirs, z, deltas, x, SR});
}

__device__ __forceinline__ float wave_excl_scan(float v, int lane, float& total) {
    float s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(s, off);
        if (lane >= off) s += t;
    }
    total = __shfl(s, 63);
    return s - v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float laplace_sigma(float s, float beta) {
    const float sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    return (1.f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
}

constexpr int MAX_CH = 4;  // SR <= 256

// ---- forward: one wave per ray ------------------------------------------------------------------
__global__ void __launch_bounds__(256) render_forward_kernel(const float* __restrict__ sdf, const uint8_t* __restrict__ valid,
                                                             const float* __restrict__ z, const float* __restrict__ deltas,
                                                             const float* __restrict__ colors, const float* __restrict__ beta_p,
                                                             int R, int SR, float* __restrict__ weights, float* __restrict__ rgb,
                                                             float* __restrict__ depth, float* __restrict__ dist, float* __restrict__ acc,
                                                             const float* __restrict__ cam_loc, const float* __restrict__ ray_dirs,
                                                             float* __restrict__ pts_rendered, const float* __restrict__ grad,
                                                             float* __restrict__ normal, const uint8_t* __restrict__ ray_valid, float depth_fill) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    const float beta = *beta_p;
    float carry = 0.f, W = 0.f, N = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    for (int base = 0; base < SR; base += 64) {
        const int s = base + lane;
        float E = 0.f, zz = 0.f;
        size_t g = (size_t)r * SR + s;
        if (s < SR) {
            zz = z[g];
            if (valid[g]) E = deltas[g] * laplace_sigma(sdf[g], beta);
        }
        float tot;
        const float excl = carry + wave_excl_scan(E, lane, tot);
        carry += tot;
        const float w = (1.f - expf(-E)) * expf(-excl);
        if (s < SR) {
            weights[g] = w;
            W += w;
            N += w * zz;
            if (colors) {
                c0 += w * colors[g * 3];
                c1 += w * colors[g * 3 + 1];
                c2 += w * colors[g * 3 + 2];
            }
            if (grad && valid[g]) {      // evaluation: normal_map = sum_j w_j g_j / |g_j| over the valid slots (pointneus_disent.py:797-815)
                const float gx = grad[g * 3], gy = grad[g * 3 + 1], gz = grad[g * 3 + 2];
                const float nrm = sqrtf((gx * gx + gy * gy) + gz * gz);
                n0 += w * (gx / nrm);
                n1 += w * (gy / nrm);
                n2 += w * (gz / nrm);
            }
        }
    }
    W = wave_sum(W);
    N = wave_sum(N);
    if (grad) {
        n0 = wave_sum(n0);
        n1 = wave_sum(n1);
        n2 = wave_sum(n2);
    }
    if (colors) {               // colors == NULL: the weights-only form (spf_render_rgb composites later, behind the colour MLPs)
        c0 = wave_sum(c0);
        c1 = wave_sum(c1);
        c2 = wave_sum(c2);
    }
    if (lane == 0) {
        if (colors) {
            rgb[3 * r] = c0;
            rgb[3 * r + 1] = c1;
            rgb[3 * r + 2] = c2;
        }
        // ray_valid (evaluation outputs): a ray without any neighbour keeps the reference's initial value of `depth` (pointneus_disent.py:822-826)
        depth[r] = (ray_valid && !ray_valid[r]) ? depth_fill : N / (W + 1e-8f);
        if (grad) {
            normal[3 * r] = n0;
            normal[3 * r + 1] = n1;
            normal[3 * r + 2] = n2;
        }
        const float dm = N / (W + 1e-10f);
        dist[r] = dm;
        acc[r] = W;
        if (pts_rendered) {       // pointneus_disent.py:765-767: the rendered surface point o + d * dist_map (multiply, then add)
#pragma unroll
            for (int c = 0; c < 3; ++c) pts_rendered[3 * r + c] = cam_loc[3 * r + c] + ray_dirs[3 * r + c] * dm;
        }
    }
}

// ---- the colour composite on its own: rgb = sum_j w_j c_j (same lane assignment and reduction tree as render_forward_kernel: the two
//      forms give the same bits), and its backward g_c = w g_rgb, g_w = c . g_rgb.  They exist so that everything that depends on the
//      WEIGHTS only (depth, dist_map, the rendered points of the pseudo-point loss and the whole pass behind them) need not wait for the
//      colour MLPs, and so that the colour branch of the backward need not wait for the pseudo-point pass's gradient.
__global__ void __launch_bounds__(256) render_rgb_kernel(const float* __restrict__ weights, const float* __restrict__ colors, int R, int SR,
                                                         float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int base = 0; base < SR; base += 64) {
        const int s = base + lane;
        if (s < SR) {
            const size_t g = (size_t)r * SR + s;
            const float w = weights[g];
            c0 += w * colors[g * 3];
            c1 += w * colors[g * 3 + 1];
            c2 += w * colors[g * 3 + 2];
        }
    }
    c0 = wave_sum(c0);
    c1 = wave_sum(c1);
    c2 = wave_sum(c2);
    if (lane == 0) {
        rgb[3 * r] = c0;
        rgb[3 * r + 1] = c1;
        rgb[3 * r + 2] = c2;
    }
}

__global__ void render_rgb_backward_kernel(const float* __restrict__ weights, const float* __restrict__ colors, const float* __restrict__ g_rgb,
                                           int R, int SR, float* __restrict__ g_colors, float* __restrict__ g_weights) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)R * SR) return;
    const int r = (int)(g / SR);
    const float gr0 = g_rgb[3 * r], gr1 = g_rgb[3 * r + 1], gr2 = g_rgb[3 * r + 2];
    const float w = weights[g];
    g_weights[g] = gr0 * colors[g * 3] + gr1 * colors[g * 3 + 1] + gr2 * colors[g * 3 + 2];      // the fused kernel's `cdot`
    g_colors[g * 3] = w * gr0;
    g_colors[g * 3 + 1] = w * gr1;
    g_colors[g * 3 + 2] = w * gr2;
}

// ---- backward ------------------------------------------------------------------------------------
// dL/dw_j = gw_j + g_rgb . c_j + g_depth (z_j/(W+e8) - N/(W+e8)^2) + g_dist (z_j/(W+e10) - N/(W+e10)^2)
// dL/dE_i = dL/dw_i e^{-E_i} T_i - sum_{j>i} dL/dw_j w_j ;  dE/dsigma = delta
// dsigma/ds = -0.5 e^{-|s|/b} / b^2 (0 at s == 0, as autograd of sign/abs) ; dsigma/db = -sigma/b + 0.5 s e^{-|s|/b} / b^3
__global__ void __launch_bounds__(256) render_backward_kernel(const float* __restrict__ sdf, const uint8_t* __restrict__ valid,
                                                              const float* __restrict__ z, const float* __restrict__ deltas,
                                                              const float* __restrict__ colors, const float* __restrict__ beta_p,
                                                              const float* __restrict__ weights, const float* __restrict__ g_weights,
                                                              const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                                              const float* __restrict__ g_dist, int R, int SR,
                                                              float* __restrict__ g_sdf, float* __restrict__ g_colors,
                                                              float* __restrict__ g_beta, const float* __restrict__ beta_param,
                                                              const float* __restrict__ g_acc, const float* __restrict__ g_pts,
                                                              const float* __restrict__ ray_dirs, long long* __restrict__ g_beta_fixed,
                                                              const int32_t* __restrict__ lfirst, const float* __restrict__ lcoef,
                                                              const float* __restrict__ lscale) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    const float beta = *beta_p;
    // g_rgb == NULL: the colour composite ran on its own (spf_render_rgb) and its backward already put c . g_rgb into g_weights
    const float gr0 = g_rgb ? g_rgb[3 * r] : 0.f, gr1 = g_rgb ? g_rgb[3 * r + 1] : 0.f, gr2 = g_rgb ? g_rgb[3 * r + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f;
    float gs = g_dist ? g_dist[r] : 0.f;
    if (g_pts)                    // pts_rendered = o + d * dist: d L / d dist += g_pts . d
        gs += (g_pts[3 * r] * ray_dirs[3 * r] + g_pts[3 * r + 1] * ray_dirs[3 * r + 1]) + g_pts[3 * r + 2] * ray_dirs[3 * r + 2];
    const float ga = g_acc ? g_acc[r] : 0.f;          // acc = sum_j w_j: the same gradient for every slot of the ray
    const int nch = (SR + 63) / 64;
    float w_[MAX_CH], z_[MAX_CH], gw_[MAX_CH];
    float W = 0.f, N = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAX_CH; ++ch) {
        w_[ch] = 0.f;
        z_[ch] = 0.f;
        gw_[ch] = 0.f;
        const int s = ch * 64 + lane;
        if (ch < nch && s < SR) {
            const size_t g = (size_t)r * SR + s;
            w_[ch] = weights[g];
            z_[ch] = z[g];
            W += w_[ch];
            N += w_[ch] * z_[ch];
        }
    }
    W = wave_sum(W);
    N = wave_sum(N);
    const float i8 = 1.f / (W + 1e-8f), i10 = 1.f / (W + 1e-10f);
    float P_ = 0.f;  // sum_j dL/dw_j w_j over all slots
#pragma unroll
    for (int ch = 0; ch < MAX_CH; ++ch) {
        const int s = ch * 64 + lane;
        if (ch < nch && s < SR) {
            const size_t g = (size_t)r * SR + s;
            const float cdot = g_rgb ? gr0 * colors[g * 3] + gr1 * colors[g * 3 + 1] + gr2 * colors[g * 3 + 2] : 0.f;
            gw_[ch] = ((g_weights ? g_weights[g] : 0.f) + ga) + cdot + gd * (z_[ch] * i8 - N * i8 * i8) + gs * (z_[ch] * i10 - N * i10 * i10);
            P_ += gw_[ch] * w_[ch];
            if (g_colors) {
                g_colors[g * 3] = w_[ch] * gr0;
                g_colors[g * 3 + 1] = w_[ch] * gr1;
                g_colors[g * 3 + 2] = w_[ch] * gr2;
            }
        }
    }
    P_ = wave_sum(P_);
    // the feature-consistency term's gradient (spf_local_forward): two entries of this ray's row
    const int lf = lfirst ? lfirst[r] : -1;
    const float lg0 = lf >= 0 ? *lscale * lcoef[2 * r] : 0.f, lg1 = lf >= 0 ? *lscale * lcoef[2 * r + 1] : 0.f;
    float carryE = 0.f, carryP = 0.f, gb = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAX_CH; ++ch) {
        if (ch >= nch) break;
        const int s = ch * 64 + lane;
        const size_t g = (size_t)r * SR + s;
        float E = 0.f, sd = 0.f, dl = 0.f, sig = 0.f;
        bool v = false;
        if (s < SR) {
            v = valid[g] != 0;
            if (v) {
                sd = sdf[g];
                dl = deltas[g];
                sig = laplace_sigma(sd, beta);
                E = dl * sig;
            }
        }
        float totE, totP;
        const float exclE = carryE + wave_excl_scan(E, lane, totE);
        const float pw = gw_[ch] * w_[ch];
        const float inclP = carryP + wave_excl_scan(pw, lane, totP) + pw;
        carryE += totE;
        carryP += totP;
        if (s < SR) {
            const float gE = gw_[ch] * expf(-E) * expf(-exclE) - (P_ - inclP);
            float gsd = 0.f;
            if (v) {
                const float eu = expf(-fabsf(sd) / beta);
                const float gsig = gE * dl;
                gsd = sd != 0.f ? gsig * (-0.5f * eu / (beta * beta)) : 0.f;
                gb += gsig * (-sig / beta + 0.5f * sd * eu / (beta * beta * beta));
            }
            if (s == lf) gsd += lg0;
            if (s == lf + 1 && lf >= 0) gsd += lg1;
            g_sdf[g] = gsd;
        }
    }
    gb = wave_sum(gb);
    if (lane == 0 && gb != 0.f) {   // beta = |beta_param| + beta_min: chain through the abs when the raw parameter is given
        if (beta_param) gb *= *beta_param > 0.f ? 1.f : (*beta_param < 0.f ? -1.f : 0.f);
        if (g_beta_fixed) fixed_add(g_beta_fixed, 0, gb);         // order-independent (common.h): one term per ray
        else atomicAdd(g_beta, gb);
    }
}

}  // namespace

extern "C" {

int spf_filter_points(const float* loc, const uint8_t* slot_valid, const float* cam_loc, const float* ray_dirs, int32_t R,
                      int32_t SR, float* z, float* deltas, float* x, void* stream) {
    if (R < 0 || SR < 1) return spf::fail(SPF_EINVAL, "spf_filter_points: bad sizes");
    if (R == 0) return SPF_OK;
    if (!loc || !slot_valid || !cam_loc || !ray_dirs || !z || !deltas || !x) return spf::fail(SPF_EINVAL, "spf_filter_points: null pointer");
    filter_points_kernel<<<spf::div_up((long long)R * SR, 256), 256, 0, (hipStream_t)stream>>>(loc, slot_valid, cam_loc, ray_dirs, R, SR, z,
                                                                                                deltas, x);
    SPF_LAUNCH_CHECK("filter_points_kernel");
    return SPF_OK;
}

int spf_render_forward(const float* sdf, const uint8_t* slot_valid, const float* z, const float* deltas, const float* colors,
                       const float* beta, int32_t R, int32_t SR, float* weights, float* rgb, float* depth, float* dist, float* acc,
                       const float* cam_loc, const float* ray_dirs, float* pts_rendered, const float* grad, float* normal, const uint8_t* ray_valid,
                       float depth_fill, void* stream) {
    if (R < 0 || SR < 1 || SR > 64 * MAX_CH) return spf::fail(SPF_EINVAL, "spf_render_forward: need 1 <= SR <= %d", 64 * MAX_CH);
    if (R == 0) return SPF_OK;
    if (!sdf || !slot_valid || !z || !deltas || !beta || !weights || !depth || !dist || !acc)
        return spf::fail(SPF_EINVAL, "spf_render_forward: null pointer");
    if ((colors == nullptr) != (rgb == nullptr)) return spf::fail(SPF_EINVAL, "spf_render_forward: colors and rgb are given (or left out) together");
    if (pts_rendered && (!cam_loc || !ray_dirs)) return spf::fail(SPF_EINVAL, "spf_render_forward: pts_rendered needs cam_loc and ray_dirs");
    if ((grad == nullptr) != (normal == nullptr)) return spf::fail(SPF_EINVAL, "spf_render_forward: grad and normal are given (or left out) together");
    render_forward_kernel<<<spf::div_up((long long)R * 64, 256), 256, 0, (hipStream_t)stream>>>(sdf, slot_valid, z, deltas, colors, beta, R,
                                                                                                 SR, weights, rgb, depth, dist, acc, cam_loc,
                                                                                                 ray_dirs, pts_rendered, grad, normal, ray_valid, depth_fill);
    SPF_LAUNCH_CHECK("render_forward_kernel");
    return SPF_OK;
}

int spf_render_rgb(const float* weights, const float* colors, int32_t R, int32_t SR, float* rgb, void* stream) {
    if (R < 0 || SR < 1 || SR > 64 * MAX_CH) return spf::fail(SPF_EINVAL, "spf_render_rgb: need 1 <= SR <= %d", 64 * MAX_CH);
    if (R == 0) return SPF_OK;
    if (!weights || !colors || !rgb) return spf::fail(SPF_EINVAL, "spf_render_rgb: null pointer");
    render_rgb_kernel<<<spf::div_up((long long)R * 64, 256), 256, 0, (hipStream_t)stream>>>(weights, colors, R, SR, rgb);
    SPF_LAUNCH_CHECK("render_rgb_kernel");
    return SPF_OK;
}

int spf_render_rgb_backward(const float* weights, const float* colors, const float* g_rgb, int32_t R, int32_t SR, float* g_colors,
                            float* g_weights, void* stream) {
    if (R < 0 || SR < 1) return spf::fail(SPF_EINVAL, "spf_render_rgb_backward: bad sizes");
    if (R == 0) return SPF_OK;
    if (!weights || !colors || !g_rgb || !g_colors || !g_weights) return spf::fail(SPF_EINVAL, "spf_render_rgb_backward: null pointer");
    render_rgb_backward_kernel<<<spf::div_up((long long)R * SR, 256), 256, 0, (hipStream_t)stream>>>(weights, colors, g_rgb, R, SR, g_colors, g_weights);
    SPF_LAUNCH_CHECK("render_rgb_backward_kernel");
    return SPF_OK;
}

int spf_render_backward(const float* sdf, const uint8_t* slot_valid, const float* z, const float* deltas, const float* colors,
                        const float* beta, const float* weights, const float* g_weights, const float* g_rgb, const float* g_depth,
                        const float* g_dist, int32_t R, int32_t SR, float* g_sdf, float* g_colors, float* g_beta, const float* beta_param,
                        const float* g_acc, const float* g_pts_rendered, const float* ray_dirs, int64_t* g_beta_fixed, const int32_t* lfirst,
                        const float* lcoef, const float* lscale, void* stream) {
    if (R < 0 || SR < 1 || SR > 64 * MAX_CH) return spf::fail(SPF_EINVAL, "spf_render_backward: need 1 <= SR <= %d", 64 * MAX_CH);
    if (R == 0) return SPF_OK;
    if (!sdf || !slot_valid || !z || !deltas || !beta || !weights || !g_sdf || !g_beta)
        return spf::fail(SPF_EINVAL, "spf_render_backward: null pointer");
    if (g_rgb && (!colors || !g_colors)) return spf::fail(SPF_EINVAL, "spf_render_backward: g_rgb needs colors and g_colors");
    if (!g_rgb && g_colors) return spf::fail(SPF_EINVAL, "spf_render_backward: g_colors needs g_rgb");
    if (g_pts_rendered && !ray_dirs) return spf::fail(SPF_EINVAL, "spf_render_backward: g_pts_rendered needs ray_dirs");
    if ((lfirst != nullptr) != (lcoef != nullptr) || (lfirst != nullptr) != (lscale != nullptr))
        return spf::fail(SPF_EINVAL, "spf_render_backward: lfirst, lcoef and lscale are given (or left out) together");
    render_backward_kernel<<<spf::div_up((long long)R * 64, 256), 256, 0, (hipStream_t)stream>>>(
        sdf, slot_valid, z, deltas, colors, beta, weights, g_weights, g_rgb, g_depth, g_dist, R, SR, g_sdf, g_colors, g_beta, beta_param, g_acc,
        g_pts_rendered, ray_dirs, reinterpret_cast<long long*>(g_beta_fixed), lfirst, lcoef, lscale);
    SPF_LAUNCH_CHECK("render_backward_kernel");
    return SPF_OK;
}

}  // extern "C"
