// Pixel -> ray set-up of one view (K9 of DESIGN.md).
//
//  spf_camera_rays   rend_util.get_camera_params + lift (spurfies/utils/rend_util.py:60-95,143-156) for the two calls
//                    the model makes per forward (pointneus_disent.py:640-650): world-space unit directions with the
//                    view's pose, and the z component of the camera-space unit direction (`depth_scale`).
//                    ~45 elementwise / bmm launches of the PyTorch formulation -> one launch.
#include "common.h"
#include "tv_body.h"
#include "color_pack.h"
#include "rhead_pack.h"

namespace {
using namespace spf;

__global__ void camera_rays_kernel(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ K, int kstride,
                                   int R, float* __restrict__ ray_dirs, float* __restrict__ cam_loc, float* __restrict__ depth_scale,
                                   const float* __restrict__ beta_param, float beta_min, float* __restrict__ beta_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0 && beta_out) *beta_out = fabsf(*beta_param) + beta_min;      // LaplaceDensity.get_beta (density.py:28-30), once per forward
    if (r >= R) return;
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[kstride + 1], cy = K[kstride + 2];
    const float x = uv[2 * r], y = uv[2 * r + 1];
    // lift(), z = 1, in the reference's operation order (rend_util.py:143-156)
    const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * 1.0f;
    const float yl = (y - cy) / fy * 1.0f;
    const float zl = 1.0f;
    const float t[3] = {pose[3], pose[7], pose[11]};
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float w = (pose[4 * i] * xl + pose[4 * i + 1] * yl + pose[4 * i + 2] * zl) + t[i];   // R c + t
        d[i] = w - t[i];                                                                           // world - cam_loc
    }
    const float n = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);                 // F.normalize, eps 1e-12
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ray_dirs[3 * r + i] = d[i] / n;
        cam_loc[3 * r + i] = t[i];
    }
    // identity pose: direction = normalize(lifted point); depth_scale = its z
    depth_scale[r] = zl / fmaxf(sqrtf(xl * xl + yl * yl + zl * zl), 1e-12f);
}

// camera_rays_kernel + sampler.hip's uniform_kernel in one launch (round 5: the first two launches of every optimisation step, 4.6 + 5.1 us
// at 128 rays): thread = (ray, sample); every thread forms its ray (the arithmetic of camera_rays_kernel, so every sample of a ray sees the
// bits the ray arrays hold), sample 0 of a ray writes the ray arrays.
// ... and, behind the ray blocks (cam_blocks of them), the per-point TV terms of the geometry latents (tv_body.h): they depend on nothing the
// step computes, so their launch (9.5 us on the critical path of a 1 ms step) rides here.
__global__ void camera_uniform_kernel(const float* __restrict__ uv, const float* __restrict__ pose, const float* __restrict__ K, int kstride,
                                      int R, float* __restrict__ ray_dirs, float* __restrict__ cam_loc, float* __restrict__ depth_scale,
                                      const float* __restrict__ beta_param, float beta_min, float* __restrict__ beta_out,
                                      const float* __restrict__ tlin, const float* __restrict__ t_rand, int n, float near, float far,
                                      float* __restrict__ z_out, float* __restrict__ points, int cam_blocks, TvArgs tv, int tv_blocks,
                                      CPackArgs ca, float* __restrict__ c_packed, float* __restrict__ c_zero, long long c_zero_floats, int cpack_blocks,
                                      RPackArgs ra, float* __restrict__ r_packed, float* __restrict__ r_zero, long long r_zero_floats, int rpack_blocks) {
    if ((int)blockIdx.x >= cam_blocks) {
        // ... and the two weight-packing jobs of the step (colour trunk, head stage: they only read the parameters) with the buffers they clear
        int b = (int)blockIdx.x - cam_blocks;
        if (b < tv_blocks) {
            tv_forward_body(tv, (long long)b * blockDim.x + threadIdx.x);
            return;
        }
        b -= tv_blocks;
        if (b < cpack_blocks) {
            color_pack_all(ca, c_packed, c_zero, c_zero_floats, (long long)b * blockDim.x + threadIdx.x, (long long)cpack_blocks * blockDim.x);
            return;
        }
        b -= cpack_blocks;
        rhead_pack_all(ra, r_packed, r_zero, r_zero_floats, (long long)b * blockDim.x + threadIdx.x, (long long)rpack_blocks * blockDim.x);
        return;
    }
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0 && beta_out) *beta_out = fabsf(*beta_param) + beta_min;
    if (gid >= (size_t)R * n) return;
    const int r = (int)(gid / n), i = (int)(gid % n);
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[kstride + 1], cy = K[kstride + 2];
    const float x = uv[2 * r], y = uv[2 * r + 1];
    const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * 1.0f;
    const float yl = (y - cy) / fy * 1.0f;
    const float zl = 1.0f;
    const float t[3] = {pose[3], pose[7], pose[11]};
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float w = (pose[4 * c] * xl + pose[4 * c + 1] * yl + pose[4 * c + 2] * zl) + t[c];
        d[c] = w - t[c];
    }
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = d[c] / nrm;
    if (i == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ray_dirs[3 * r + c] = d[c];
            cam_loc[3 * r + c] = t[c];
        }
        depth_scale[r] = zl / fmaxf(sqrtf(xl * xl + yl * yl + zl * zl), 1e-12f);
    }
    // UniformSampler.get_z_vals (ray_sampler.py:33-59), as sampler.hip's uniform_kernel
    auto zlin = [&](int k) { return near * (1.0f - tlin[k]) + far * tlin[k]; };
    float z = zlin(i);
    if (t_rand) {
        const float up = i + 1 < n ? 0.5f * (zlin(i + 1) + z) : z;
        const float lo = i > 0 ? 0.5f * (z + zlin(i - 1)) : z;
        z = lo + (up - lo) * t_rand[gid];
    }
    z_out[gid] = z;
#pragma unroll
    for (int c = 0; c < 3; ++c) points[gid * 3 + c] = t[c] + z * d[c];
}

}  // namespace

extern "C" {

int spf_camera_uniform(const float* uv, const float* pose, const float* intrinsics, int32_t k_stride, int32_t R, float* ray_dirs, float* cam_loc,
                       float* depth_scale, const float* beta_param, float beta_min, float* beta_out, const float* tlin, const float* t_rand,
                       int32_t n, float near, float far, float* z, float* points, const float* tv_feat, const int32_t* tv_nbr, const float* tv_w,
                       const float* tv_norm, int32_t tv_n, int32_t tv_k, float* tv_out, const spf_prologue_packs* packs, void* stream) {
    if (R < 0 || n < 1 || (k_stride != 3 && k_stride != 4)) return spf::fail(SPF_EINVAL, "spf_camera_uniform: need R >= 0, n >= 1, k_stride 3 or 4");
    if (R == 0) return SPF_OK;
    if (!uv || !pose || !intrinsics || !ray_dirs || !cam_loc || !depth_scale || !tlin || !z || !points) return spf::fail(SPF_EINVAL, "spf_camera_uniform: null pointer");
    if (beta_out && !beta_param) return spf::fail(SPF_EINVAL, "spf_camera_uniform: beta_out needs beta_param");
    spf::TvArgs tv{};
    int tv_blocks = 0;
    if (tv_feat) {
        if (!tv_nbr || !tv_w || !tv_norm || !tv_out || tv_n < 0 || tv_k < 1) return spf::fail(SPF_EINVAL, "spf_camera_uniform: incomplete TV arguments");
        tv = spf::TvArgs{tv_feat, tv_nbr, tv_w, tv_norm, tv_n, tv_k, tv_out, nullptr};
        tv_blocks = spf::div_up((long long)tv_n * 32, 256);
    }
    spf_prologue_packs pk{};
    int cpack_blocks = 0, rpack_blocks = 0;
    if (packs) {
        pk = *packs;
        if (!pk.cw0 || !pk.cb0 || !pk.cw2 || !pk.cb2 || !pk.cw4 || !pk.cb4 || !pk.c_packed || !pk.rw6 || !pk.rb6 || !pk.rw0 || !pk.rb0 || !pk.rw2 || !pk.rb2 ||
            !pk.rw4 || !pk.rb4 || !pk.r_packed)
            return spf::fail(SPF_EINVAL, "spf_camera_uniform: incomplete weight-packing arguments");
        if (pk.c_zero_floats < 0 || pk.r_zero_floats < 0 || (pk.c_zero && ((uintptr_t)pk.c_zero & 15)) || (pk.r_zero && ((uintptr_t)pk.r_zero & 15)))
            return spf::fail(SPF_EINVAL, "spf_camera_uniform: the buffers to clear must be 16-byte aligned");
        cpack_blocks = spf::div_up(C_PACK_THREADS, 256);
        rpack_blocks = spf::div_up(R_PACK_THREADS, 256);
    }
    const int cam_blocks = spf::div_up((long long)R * n, 256);
    camera_uniform_kernel<<<cam_blocks + tv_blocks + cpack_blocks + rpack_blocks, 256, 0, (hipStream_t)stream>>>(
        uv, pose, intrinsics, k_stride, R, ray_dirs, cam_loc, depth_scale, beta_param, beta_min, beta_out, tlin, t_rand, n, near, far, z, points, cam_blocks, tv,
        tv_blocks, CPackArgs{pk.cw0, pk.cb0, pk.cw2, pk.cb2, pk.cw4, pk.cb4}, pk.c_packed, pk.c_zero, (long long)pk.c_zero_floats, cpack_blocks,
        RPackArgs{pk.rw6, pk.rb6, pk.rw0, pk.rb0, pk.rw2, pk.rb2, pk.rw4, pk.rb4}, pk.r_packed, pk.r_zero, (long long)pk.r_zero_floats, rpack_blocks);
    SPF_LAUNCH_CHECK("camera_uniform_kernel");
    return SPF_OK;
}

int spf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int32_t k_stride, int32_t R, float* ray_dirs,
                    float* cam_loc, float* depth_scale, const float* beta_param, float beta_min, float* beta_out, void* stream) {
    if (R < 0 || (k_stride != 3 && k_stride != 4)) return spf::fail(SPF_EINVAL, "spf_camera_rays: k_stride must be 3 or 4 (got %d)", k_stride);
    if (R == 0) return SPF_OK;
    if (!uv || !pose || !intrinsics || !ray_dirs || !cam_loc || !depth_scale) return spf::fail(SPF_EINVAL, "spf_camera_rays: null pointer");
    if (beta_out && !beta_param) return spf::fail(SPF_EINVAL, "spf_camera_rays: beta_out needs beta_param");
    camera_rays_kernel<<<spf::div_up(R, 256), 256, 0, (hipStream_t)stream>>>(uv, pose, intrinsics, k_stride, R, ray_dirs, cam_loc, depth_scale,
                                                                             beta_param, beta_min, beta_out);
    SPF_LAUNCH_CHECK("camera_rays_kernel");
    return SPF_OK;
}

}  // extern "C"
