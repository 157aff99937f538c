// Multi-view feature-consistency ("local") term of the DTU recipe (K11 of DESIGN.md), one wavefront per ray.
//
//  spf_local_forward    find_surface_points            spurfies/model/pointneus_disent.py:586-612
//                       surface point o + d t          spurfies/model/pointneus_disent.py:744-749
//                       get_local_loss                 spurfies/feat_utils.py:377-451 (projection :43-55, normalisation :58-77,
//                                                      F.grid_sample bilinear / zeros / align_corners=False, cosine term, masks)
//  spf_local_backward   the autograd backward of all of it, as a dense [R,SR] gradient of the SDF rows
//
// The reference runs this as ~60 PyTorch launches with three host synchronisations (`mask.sum() == 0`, boolean indexing, `.tolist()`).
// Here every ray stays in place: a ray without a + -> - crossing contributes nothing and is not counted.
//
// The only differentiable inputs are the two SDF values of the crossing (feature maps, cameras and sample depths carry no gradient), and they
// enter through ONE scalar, the interpolated depth t.  The forward therefore carries d/dt alongside every value (forward-mode: the tangent of
// the surface point is the ray direction) and leaves per ray {crossing slot, d sum/d sdf[slot], d sum/d sdf[slot + 1]}: the backward is two
// multiply-adds per ray and rides in the compositing backward (spf_render_backward's `lfirst / lcoef / lscale`), no launch of its own.
#include "common.h"

namespace {
using namespace spf;

constexpr int MAXV = SPF_LOCAL_MAX_VIEWS;
static_assert(sizeof(spf_local_desc) == 8 * MAXV + 128 * MAXV + 32, "spf_local_desc is uploaded as raw bytes (spurfies_amd/_lib.py:LocalDesc)");
constexpr float SDF_FILL = 1000.0f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// World point `pw` (tangent `dpw`) -> unnormalised sampling position (ix, iy) in a [Hf, Wf] feature map and its tangent; in_range as
// feat_utils.get_in_range on the normalised coordinates.  cam = {extrinsic 4x4, K in [1][:3][:3]} row-major.
struct Proj {
    float ix, iy, dix, diy;
    bool in_range;
};
__device__ __forceinline__ Proj project(const float* __restrict__ cam, const float pw[3], const float dpw[3], float Wf, float Hf) {
    const float* E = cam;
    const float* K = cam + 16;
    float h[4], dh[4];
    {   // idx_world2cam: E [pw, 1], divided by (its last component + 1e-9)
        float c[4], dc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = ((E[4 * i] * pw[0] + E[4 * i + 1] * pw[1]) + E[4 * i + 2] * pw[2]) + E[4 * i + 3];
            dc[i] = (E[4 * i] * dpw[0] + E[4 * i + 1] * dpw[1]) + E[4 * i + 2] * dpw[2];
        }
        const float q = c[3] + 1e-9f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = c[i] / q;
            dh[i] = (dc[i] - h[i] * dc[3]) / q;
        }
    }
    // idx_cam2img: xyz / (w + 1e-9), K, divided by (z + 1e-9)
    const float q2 = h[3] + 1e-9f;
    float ic[3], dic[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ic[i] = h[i] / q2;
        dic[i] = (dh[i] - ic[i] * dh[3]) / q2;
    }
    float ih[3], dih[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ih[i] = (K[4 * i] * ic[0] + K[4 * i + 1] * ic[1]) + K[4 * i + 2] * ic[2];
        dih[i] = (K[4 * i] * dic[0] + K[4 * i + 1] * dic[1]) + K[4 * i + 2] * dic[2];
    }
    const float q3 = ih[2] + 1e-9f;
    const float gx = ih[0] / q3, gy = ih[1] / q3;
    const float dgx = (dih[0] - gx * dih[2]) / q3, dgy = (dih[1] - gy * dih[2]) / q3;
    // normalize_for_grid_sample(grid / 2): the maps are at half the image resolution; clamp to +-1.1
    const float nx = fminf(fmaxf((gx / 2.0f) / Wf * 2.0f - 1.0f, -1.1f), 1.1f);
    const float ny = fminf(fmaxf((gy / 2.0f) / Hf * 2.0f - 1.0f, -1.1f), 1.1f);
    Proj p;
    p.in_range = nx <= 1.0f && nx >= -1.0f && ny <= 1.0f && ny >= -1.0f;        // false for NaN
    // grid_sampler_unnormalize, align_corners = False;  d ix / d gx = (1/2)(2/Wf)(Wf/2) = 1/2 inside the clamp (in_range implies that)
    p.ix = ((nx + 1.0f) * Wf - 1.0f) / 2.0f;
    p.iy = ((ny + 1.0f) * Hf - 1.0f) / 2.0f;
    p.dix = 0.5f * dgx;
    p.diy = 0.5f * dgy;
    return p;
}

__global__ void __launch_bounds__(256)
local_forward_kernel(const spf_local_desc* __restrict__ desc, const float* __restrict__ sdf, const float* __restrict__ z,
                     const float* __restrict__ cam_loc, const float* __restrict__ ray_dirs, int R, int SR, float* __restrict__ d_surface,
                     int32_t* __restrict__ lfirst, float* __restrict__ lsum, float* __restrict__ lcoef) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    // ---- find_surface_points: the first adjacent pair of slots with sdf going from + to - (slots without a point hold 1000) ----------
    int first = -1;
    for (int base = 0; base < SR - 1 && first < 0; base += 64) {
        const int s = base + lane;
        bool cross = false;
        if (s < SR - 1) {
            const float s0 = sdf[(size_t)r * SR + s], s1 = sdf[(size_t)r * SR + s + 1];
            cross = s0 != SDF_FILL && s1 != SDF_FILL && s1 * s0 < 0.f && s1 < s0;
        }
        const unsigned long long m = __ballot(cross);
        if (m) first = base + __ffsll((long long)m) - 1;
    }
    if (first < 0) {
        if (lane == 0) {
            d_surface[r] = 0.f;
            lfirst[r] = -1;
            lsum[r] = 0.f;
            lcoef[2 * r] = 0.f;
            lcoef[2 * r + 1] = 0.f;
        }
        return;
    }
    const size_t g0 = (size_t)r * SR + first;
    const float a = sdf[g0], b = sdf[g0 + 1], d0 = z[g0], d1 = z[g0 + 1];
    const float t = (a * d1 - b * d0) / (a - b);
    // ---- surface point (tangent w.r.t. t = the ray direction), de-normalised world frame p / 2 * size + center --------------------
    const int n_src = desc->n_src, C = desc->C, H = desc->H, W = desc->W;
    const float size = desc->size;
    float pw[3], dpw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float dir = ray_dirs[3 * r + c];
        const float p = cam_loc[3 * r + c] + dir * t;
        pw[c] = p / 2.0f * size + desc->center[c];
        dpw[c] = dir / 2.0f * size;
    }
    Proj pr[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v)
        if (v <= n_src) pr[v] = project(&desc->cam[v][0][0], pw, dpw, (float)W, (float)H);
    // ---- bilinear taps: lane = channel; per view the sampled feature g and its tangent dg; partial sums over channels ------------------
    float nn[MAXV], gd[MAXV], dot[MAXV], crs[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) nn[v] = gd[v] = dot[v] = crs[v] = 0.f;
    const size_t plane = (size_t)H * W;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float gv[MAXV], dgv[MAXV];
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            gv[v] = dgv[v] = 0.f;
            if (v <= n_src && pr[v].in_range && c < C) {
                const float* __restrict__ f = reinterpret_cast<const float*>(desc->feat[v]) + (size_t)c * plane;
                const float ix = pr[v].ix, iy = pr[v].iy;
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
                const float wx1 = ix - fx0, wx0 = (fx0 + 1.0f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.0f) - iy;
                const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
                const float nw = (bx0 && by0) ? f[(size_t)y0 * W + x0] : 0.f, ne = (bx1 && by0) ? f[(size_t)y0 * W + x1] : 0.f;
                const float sw = (bx0 && by1) ? f[(size_t)y1 * W + x0] : 0.f, se = (bx1 && by1) ? f[(size_t)y1 * W + x1] : 0.f;
                gv[v] = ((nw * (wx0 * wy0) + ne * (wx1 * wy0)) + sw * (wx0 * wy1)) + se * (wx1 * wy1);
                const float gix = (ne - nw) * wy0 + (se - sw) * wy1, giy = (sw - nw) * wx0 + (se - ne) * wx1;
                dgv[v] = gix * pr[v].dix + giy * pr[v].diy;
            }
        }
#pragma unroll
        for (int v = 0; v < MAXV; ++v)
            if (v <= n_src) {
                nn[v] += gv[v] * gv[v];
                gd[v] += gv[v] * dgv[v];
                if (v > 0) {
                    dot[v] += gv[0] * gv[v];
                    crs[v] += dgv[0] * gv[v] + gv[0] * dgv[v];
                }
            }
    }
    // ---- cosine term per source view (feat_utils.py:425-437) and its tangent ----------------------------------------------------
    float sum = 0.f, dsum = 0.f;
    float n0 = 0.f, N0 = 1.f, r0 = 0.f;
#pragma unroll
    for (int v = 0; v < MAXV; ++v)
        if (v <= n_src) {
            const float n = sqrtf(wave_sum(nn[v])), N = fmaxf(n, 1e-9f);
            const float gdv = wave_sum(gd[v]);
            const float rel = n >= 1e-9f ? gdv / (n * N) : 0.f;         // d N / dt / N (the clamp passes the gradient at or above its minimum)
            if (v == 0) {
                n0 = n, N0 = N, r0 = rel;
                continue;
            }
            const float dt = wave_sum(dot[v]), cr = wave_sum(crs[v]);
            const float corr = dt / N0 / N;
            const float cl = fabsf(1.0f - corr);
            const bool keep = pr[0].in_range && pr[v].in_range && cl < 0.5f;
            if (keep) {
                const float dcorr = cr / N0 / N - corr * (r0 + rel);
                const float e = 1.0f - corr;
                sum += cl;
                dsum += (e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f)) * dcorr;
            }
        }
    (void)n0;
    if (lane == 0) {
        const float inv = 1.0f / ((a - b) * (a - b));
        d_surface[r] = t;
        lfirst[r] = first;
        lsum[r] = sum;
        lcoef[2 * r] = dsum * (b * (d0 - d1)) * inv;           // d t / d a = b (d0 - d1) / (a - b)^2
        lcoef[2 * r + 1] = dsum * (a * (d1 - d0)) * inv;       // d t / d b = a (d1 - d0) / (a - b)^2
    }
}

// dense gradient of the SDF rows: g_sdf[r, lfirst[r] + {0, 1}] = g_sum * lcoef[r, {0, 1}], zero elsewhere
__global__ void local_backward_kernel(const int32_t* __restrict__ lfirst, const float* __restrict__ lcoef, const float* __restrict__ g_sum,
                                      int R, int SR, float* __restrict__ g_sdf) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)R * SR) return;
    const int r = (int)(gid / SR), s = (int)(gid - (size_t)r * SR);
    const int f = lfirst[r];
    float g = 0.f;
    if (f >= 0 && (s == f || s == f + 1)) g = *g_sum * lcoef[2 * r + (s - f)];
    g_sdf[gid] = g;
}

}  // namespace

extern "C" {

int spf_local_forward(const spf_local_desc* desc, const float* sdf, const float* z, const float* cam_loc, const float* ray_dirs, int32_t R,
                      int32_t SR, float* d_surface, int32_t* lfirst, float* lsum, float* lcoef, void* stream) {
    if (R < 0 || SR < 2) return spf::fail(SPF_EINVAL, "spf_local_forward: need R >= 0, SR >= 2");
    if (R == 0) return SPF_OK;
    if (!desc || !sdf || !z || !cam_loc || !ray_dirs || !d_surface || !lfirst || !lsum || !lcoef)
        return spf::fail(SPF_EINVAL, "spf_local_forward: null pointer");
    local_forward_kernel<<<spf::div_up((long long)R * 64, 256), 256, 0, (hipStream_t)stream>>>(desc, sdf, z, cam_loc, ray_dirs, R, SR, d_surface,
                                                                                                lfirst, lsum, lcoef);
    SPF_LAUNCH_CHECK("local_forward_kernel");
    return SPF_OK;
}

int spf_local_backward(const int32_t* lfirst, const float* lcoef, const float* g_sum, int32_t R, int32_t SR, float* g_sdf, void* stream) {
    if (R < 0 || SR < 2) return spf::fail(SPF_EINVAL, "spf_local_backward: need R >= 0, SR >= 2");
    if (R == 0) return SPF_OK;
    if (!lfirst || !lcoef || !g_sum || !g_sdf) return spf::fail(SPF_EINVAL, "spf_local_backward: null pointer");
    local_backward_kernel<<<spf::div_up((long long)R * SR, 256), 256, 0, (hipStream_t)stream>>>(lfirst, lcoef, g_sum, R, SR, g_sdf);
    SPF_LAUNCH_CHECK("local_backward_kernel");
    return SPF_OK;
}

}  // extern "C"
