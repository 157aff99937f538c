// VolSDF error-bounded sampler (K7 of DESIGN.md): one wavefront per ray, every per-ray cumulative sum a
// wave prefix-sum over a blocked register layout, CDF inversion by binary search in LDS.
//
// Replaces the ~150 elementwise / cumsum / gather / sort launches per call of
//   UniformSampler.get_z_vals        spurfies/model/ray_sampler.py:33-59
//   ErrorBoundSampler_pn.get_z_vals  spurfies/model/ray_sampler.py:377-574
//   ErrorBoundSampler_pn.get_error_bound  :576-588
// The SDF evaluations between the stages (model.sdf_importance, :403) are the kNN + geometry kernels; the
// random numbers are drawn by the host from the CPU generator exactly as the reference does and handed in.
#include <cfloat>

#include "common.h"
#include "grid_dev.h"

namespace {
using namespace spf;

__device__ __forceinline__ float sigma_laplace(float s, float beta) {
    const float sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    return (1.f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// torch.max propagates NaN (rays that miss everything carry NaN samples in the reference too); fmaxf would drop it
__device__ __forceinline__ float nanmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nanmax(v, __shfl_xor(v, off));
    return v;
}
// ordering key: NaN sorts last, as torch.sort does
__device__ __forceinline__ float okey(float v) { return v != v ? __builtin_inff() : v; }
__device__ __forceinline__ float wexcl(float v, int lane) {  // exclusive scan of one value per lane
    float s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(s, off);
        if (lane >= off) s += t;
    }
    return s - v;
}

// ---- stage 0: stratified uniform samples + their 3-D points ---------------------------------------
__global__ void uniform_kernel(const float* __restrict__ tlin, const float* __restrict__ t_rand, const float* __restrict__ cam_loc,
                               const float* __restrict__ ray_dirs, int R, int n, float near, float far, float* __restrict__ z_out,
                               float* __restrict__ points) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)R * n) return;
    const int r = (int)(gid / n), i = (int)(gid % n);
    auto zl = [&](int k) { return near * (1.0f - tlin[k]) + far * tlin[k]; };
    float z = zl(i);
    if (t_rand) {
        const float up = i + 1 < n ? 0.5f * (zl(i + 1) + z) : z;
        const float lo = i > 0 ? 0.5f * (z + zl(i - 1)) : z;
        z = lo + (up - lo) * t_rand[gid];
    }
    z_out[gid] = z;
#pragma unroll
    for (int c = 0; c < 3; ++c) points[gid * 3 + c] = cam_loc[3 * r + c] + z * ray_dirs[3 * r + c];
}

// Round 5, the optimisation step's sampler pass (fast = 1: ONE iteration that goes straight to the final sampling) as one launch instead of
// four: FUSED adds, in front, the per-point reduction of the geometry kernel's per-pair scratch (geo_point_reduce_kernel's sums, same order:
// sdf = sum_j w_j sdf_j / sum_j w_j over the sample's pairs, 1000 where the sample has no neighbour) and, behind the inverse CDF, the work of
// sampler_finish_kernel (rank sort of [samples | near | far | z[sel]], the main-pass points o + z d) and of grid.hip's hit_slots_kernel (the
// ray's first SR samples inside the dilated occupancy get slots 0..): the chain reduce -> iterate -> finish -> hit_slots was 4 launches of
// 5 - 18 us each on the critical path of a 1 ms step (128 rays per GPU).
struct SamplerTail {
    const float* pair_tmp;        // [pairs, GEO_PT_STRIDE] = {w, sdf_j, ...} from spf_geo_forward(sdf = NULL)
    const int32_t* pair_off;      // [points + 1]
    const int32_t* slot_point;    // [R * n]: point id of a sample or -1
    const int32_t* sel;           // [Ne] extra samples of the current z
    int Ne;
    float near, far;
    const float* cam_loc;
    const float* ray_dirs;
    float* z_out;                 // [R, N + 2 + Ne]
    float* points;                // [R, N + 2 + Ne, 3]
    GridDev grid;
    int SR;
    int32_t* slot_sample;         // [R, SR]
    uint8_t* ray_valid;           // [R] (cleared; the kNN kernel raises it)
    // ---- evaluation loop (spf_sampler_eval): the iteration's SDF row is gathered from the previous row and the NEW samples' pair scratch
    const float* sdf_prev;        // [R, n_prev] SDF of the previous iteration's z (NULL: first iteration)
    const int32_t* merged_idx;    // [R, n] index of z[k] in cat(previous z, new samples) (NULL: identity)
    int n_prev;                   // n - (number of new samples); pair_off / slot_point describe the new samples only
    float* sdf_cur;               // [R, n] out: this iteration's SDF row (the next launch / iteration reads it)
    const float* u_alt;           // MODE_EVAL_STEP: u / N of the FINAL branch (the kernel's u / N are the merging branch's)
    int N_alt;
    float* points_new;            // [R, N, 3] o + s d of the merging branch's new samples (the next iteration's query points)
};

// MODE: what surrounds the per-ray arithmetic
constexpr int MODE_PLAIN = 0;       // sdf_in -> samples / merge (spf_sampler_iter)
constexpr int MODE_TRAIN = 1;       // pair scratch -> final samples -> finish + hit slots (spf_sampler_train)
constexpr int MODE_EVAL_TEST = 2;   // pair scratch + previous row -> sdf_cur, beta, convergence flag; nothing else
constexpr int MODE_EVAL_STEP = 3;   // sdf_cur + beta; flags[it + 1] decides: merge (+ new points) | final samples -> finish + hit slots
constexpr int MODE_EVAL_LAST = 4;   // pair scratch + previous row -> full bisection -> final samples -> finish + hit slots

template <int E, int MODE = MODE_PLAIN>
__global__ void __launch_bounds__(256) sampler_iter_kernel(const float* __restrict__ z_in, const float* __restrict__ sdf_in,
                                                           const float* __restrict__ beta_in, const float* __restrict__ beta0_p, int R, int n,
                                                           float eps, float bound_coef, int beta_iters, int more, float add_tiny,
                                                           const float* __restrict__ u, int u_per_ray, int N, float* __restrict__ samples,
                                                           float* __restrict__ beta_out, float* __restrict__ z_merged,
                                                           int32_t* __restrict__ merged_idx, int32_t* __restrict__ flags, int it,
                                                           SamplerTail tail) {
    // Device-side loop control (evaluation mode without a host sync per iteration): flags[i] != 0 <=> the sampler loop reaches iteration i.
    // Every pass of iteration `it` needs flags[it]; the sampling passes are additionally tied to the convergence test of the SAME iteration,
    // flags[it + 1] ("beta.max() > beta0", ray_sampler.py:468, set below by the beta-only pass): the merging pass runs iff it is set, the final
    // pass iff it is clear.  A pass whose turn it is not returns at once.
    constexpr bool FUSED = MODE == MODE_TRAIN || MODE == MODE_EVAL_TEST || MODE == MODE_EVAL_LAST;     // the input stage reads the pair scratch
    if (MODE == MODE_EVAL_STEP) {
        const bool reached = flags[it] != 0;
        more = reached && flags[it + 1] != 0;
        if (!more) {
            // no further iteration (the loop ends here, or never got here): the NEXT iteration's neighbour search still runs over points_new
            // (only its MLP work is gated off).  NaN is outside every cell, so that search costs nothing — stale points of an earlier
            // replay would be searched in full.
            const int r0 = blockIdx.x * 4 + (threadIdx.x >> 6);
            if (tail.points_new && r0 < R)
                for (int m = (threadIdx.x & 63); m < 3 * N; m += 64) tail.points_new[(size_t)r0 * 3 * N + m] = __builtin_nanf("");
            if (!reached) return;
            u = tail.u_alt;
            N = tail.N_alt;
        }
    } else if (flags) {
        bool live = flags[it] != 0;
        if (N > 0) live = live && ((flags[it + 1] != 0) == (more != 0));
        if (!live) return;
    }
    const bool tailf = MODE == MODE_TRAIN || MODE == MODE_EVAL_LAST || (MODE == MODE_EVAL_STEP && !more);
    constexpr int NMAX = 64 * E;
    __shared__ float smem[4 * (3 * NMAX + 128)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    const bool active = r < R;
    float* zs = smem + wave * (3 * NMAX + 128);
    float* ds = zs + NMAX;
    float* cdf = ds + NMAX;
    float* sm = cdf + NMAX;  // [128] new samples (merge)
    if (active)
        for (int k = lane; k < n; k += 64) {
            zs[k] = z_in[(size_t)r * n + k];
            if (FUSED) {          // geo_point_reduce_kernel's weighted mean of this sample's pairs (same order of additions)
                // evaluation loop: z[k] is either a sample of an earlier iteration (its SDF is in the previous row) or one of this
                // iteration's n - n_prev new samples (ray_sampler.py:405-410: gather(cat(sdf, new), samples_idx))
                const int src = tail.merged_idx ? tail.merged_idx[(size_t)r * n + k] : k;
                const int p = src < tail.n_prev ? -2 : tail.slot_point[(size_t)r * (n - tail.n_prev) + (src - tail.n_prev)];
                float v = 1000.0f;                                    // pointneus_disent.py:371 filler
                if (p == -2) v = tail.sdf_prev[(size_t)r * tail.n_prev + src];
                if (p >= 0) {
                    const int q0 = tail.pair_off[p], q1 = tail.pair_off[p + 1];
                    float nrm = 0.f, acc = 0.f;
                    for (int q = q0; q < q1; ++q) {
                        const float w = tail.pair_tmp[(size_t)q * GEO_PT_STRIDE];
                        nrm += w;
                        acc += w * tail.pair_tmp[(size_t)q * GEO_PT_STRIDE + 1];
                    }
                    v = acc / nrm;
                }
                ds[k] = v;
                if (tail.sdf_cur) tail.sdf_cur[(size_t)r * n + k] = v;
            } else {
                ds[k] = sdf_in[(size_t)r * n + k];
            }
        }
    __syncthreads();
    const float beta0 = *beta0_p;
    float a[E], dst[E], sd[E];
    float sumsq = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        a[e] = 0.f;
        dst[e] = 0.f;
        sd[e] = 0.f;
        if (active && i < n) sd[e] = ds[i];
        if (active && i < n - 1) {
            const float d0 = ds[i], d1 = ds[i + 1];
            const float aa = zs[i + 1] - zs[i], b = fabsf(d0), c = fabsf(d1);
            a[e] = aa;
            sumsq += aa * aa;
            const bool first = aa * aa + b * b <= c * c, second = aa * aa + c * c <= b * b;
            float v = 0.f;
            if (!first && !second && (b + c - aa > 0.f)) {
                const float s = (aa + b + c) / 2.0f;
                v = (2.0f * sqrtf(s * (s - aa) * (s - b) * (s - c))) / aa;
            }
            if (first) v = b;
            if (second) v = c;
            const float s0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f), s1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
            dst[e] = (s1 * s0 == 1.f) ? v : 0.f;
        }
    }
    // error bound of ray_sampler.py:576-588 for one beta (uniform over the wave)
    auto errbound = [&](float beta, float* T_out, float* EI_out) {
        float ep_inc[E], fe_exc[E];
        float run_ep = 0.f, run_fe = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            float ep = 0.f, fe = 0.f;
            if (i < n - 1) {
                ep = expf(-dst[e] / beta) * (a[e] * a[e]) / (4.f * (beta * beta));
                fe = a[e] * sigma_laplace(sd[e], beta);
            }
            run_ep += ep;
            ep_inc[e] = run_ep;
            fe_exc[e] = run_fe;
            run_fe += fe;
        }
        const float off_ep = wexcl(run_ep, lane), off_fe = wexcl(run_fe, lane);
        float mx = -FLT_MAX;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            const float ei = off_ep + ep_inc[e], tr = expf(-(off_fe + fe_exc[e]));
            if (T_out) {
                T_out[e] = tr;
                EI_out[e] = ei;
            }
            if (i < n - 1) mx = nanmax(mx, (fminf(expf(ei), 1.0e6f) - 1.0f) * tr);
        }
        return wmax(mx);
    };
    float beta = beta_in ? (active ? beta_in[r] : 1.f) : sqrtf(bound_coef * wsum(sumsq));
    if (errbound(beta0, nullptr, nullptr) <= eps) beta = beta0;
    float bmin = beta0, bmax = beta;
    for (int it = 0; it < beta_iters; ++it) {
        const float mid = (bmin + bmax) / 2.0f;
        const float err = errbound(mid, nullptr, nullptr);
        if (err <= eps) bmax = mid;
        if (err > eps) bmin = mid;
    }
    beta = bmax;
    if (active && lane == 0) {
        beta_out[r] = beta;
        if (flags && N == 0 && beta > beta0) atomicOr(&flags[it + 1], 1);      // the beta-only pass is the convergence test: another iteration follows
    }
    if (MODE == MODE_EVAL_TEST) return;

    // ---- pdf over the n-1 intervals with the final beta -------------------------------------------
    float p[E];
    if (more) {  // proportional to the current error bound (:470-489)
        float T[E], EI[E];
        errbound(beta, T, EI);
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = (lane * E + e < n - 1) ? (fminf(expf(EI[e]), 1.0e6f) - 1.0f) * T[e] + add_tiny : 0.f;
    } else {     // proportional to the rendering weights (:447-464, 491-503)
        float fe[E], run = 0.f, ex[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            fe[e] = 0.f;
            if (i < n) fe[e] = (i < n - 1 ? a[e] : 1e10f) * sigma_laplace(sd[e], beta);
            ex[e] = run;
            run += fe[e];
        }
        const float off = wexcl(run, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float w = (1.f - expf(-fe[e])) * expf(-(off + ex[e]));
            p[e] = (lane * E + e < n - 1) ? w + 1e-5f : 0.f;
        }
    }
    float tot = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) tot += p[e];
    tot = wsum(tot);
    {
        float run = 0.f, ex[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            p[e] = p[e] / tot;
            ex[e] = run;
            run += p[e];
        }
        const float off = wexcl(run, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            if (i < n) cdf[i] = off + ex[e];   // cdf[k] = sum_{i<k} pdf_i, cdf[0] = 0
        }
    }
    __syncthreads();
    // ---- inverse CDF (:517-529) ------------------------------------------------------------------
    if (active)
        for (int m = lane; m < N; m += 64) {
            const float uu = u_per_ray ? u[(size_t)r * N + m] : u[m];
            int lo = 0, hi = n;
            while (lo < hi) {           // searchsorted(right=True): number of cdf entries <= uu
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
            }
            const int below = max(lo - 1, 0), above = min(lo, n - 1);
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.f;
            const float t = (uu - cdf[below]) / denom;
            const float s = zs[below] + t * (zs[above] - zs[below]);
            if (MODE == MODE_PLAIN) samples[(size_t)r * N + m] = s;
            if (more || tailf) sm[m] = s;
            if (MODE == MODE_EVAL_STEP && more) {       // the next iteration's query points (ray_sampler.py:401-402)
#pragma unroll
                for (int k = 0; k < 3; ++k) tail.points_new[((size_t)r * N + m) * 3 + k] = tail.cam_loc[3 * r + k] + s * tail.ray_dirs[3 * r + k];
            }
        }
    if (tailf) {
        // ---- sampler_finish_kernel: z_final = sort([samples | near | far | z[sel]]) by rank counting, points = o + z d -----------------
        __syncthreads();                                           // cdf is dead from here on: its LDS holds the candidates
        const int M = N + 2 + tail.Ne;
        float* c = cdf;                                            // [<= 128] candidates (M <= 128 is checked by the host)
        float* srt = ds;                                           // [<= 128] sorted (the SDF values are in registers)
        if (active)
            for (int q = lane; q < M; q += 64) {
                float v;
                if (q < N) v = sm[q];
                else if (q == N) v = tail.near;
                else if (q == N + 1) v = tail.far;
                else v = zs[tail.sel[q - N - 2]];
                c[q] = v;
            }
        __syncthreads();
        float o3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f};
        if (active) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o3[k] = tail.cam_loc[3 * r + k];
                d3[k] = tail.ray_dirs[3 * r + k];
            }
            for (int q = lane; q < M; q += 64) {
                const float v = c[q], kv = okey(v);
                int rank = 0;
                for (int pth = 0; pth < M; ++pth) {
                    const float w = okey(c[pth]);
                    rank += (w < kv) || (w == kv && pth < q);
                }
                const size_t o = (size_t)r * M + rank;
                tail.z_out[o] = v;
                srt[rank] = v;
#pragma unroll
                for (int k = 0; k < 3; ++k) tail.points[o * 3 + k] = o3[k] + v * d3[k];
            }
        }
        __syncthreads();
        // ---- hit_slots_kernel: samples are tested 64 at a time in depth order, hits get consecutive slots ----------------------------
        if (active) {
            const int SR = tail.SR;
            if (lane == 0) tail.ray_valid[r] = 0;
            int count = 0;
            for (int base = 0; base < M && count < SR; base += 64) {
                const int d = base + lane;
                bool hit = false;
                if (d < M) {
                    const float v = srt[d];
                    hit = dil_hit(tail.grid, o3[0] + v * d3[0], o3[1] + v * d3[1], o3[2] + v * d3[2]);
                }
                const unsigned long long b = __ballot(hit);
                const int slot = count + __popcll(b & ((1ull << lane) - 1ull));
                if (hit && slot < SR) tail.slot_sample[(size_t)r * SR + slot] = d;
                count += __popcll(b);
            }
            for (int sl = min(count, SR) + lane; sl < SR; sl += 64) tail.slot_sample[(size_t)r * SR + sl] = -1;
        }
        return;
    }
    if (more) {  // z_new = sort(cat(z, samples)) as a merge of two sorted lists (:532-533)
        __syncthreads();
        if (active) {
            for (int i = lane; i < n; i += 64) {
                const float v = zs[i], kv = okey(v);
                int lo = 0, hi = N;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (okey(sm[mid]) < kv) lo = mid + 1; else hi = mid; }
                z_merged[(size_t)r * (n + N) + i + lo] = v;
                merged_idx[(size_t)r * (n + N) + i + lo] = i;
            }
            for (int m = lane; m < N; m += 64) {
                const float v = sm[m], kv = okey(v);
                int lo = 0, hi = n;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (okey(zs[mid]) <= kv) lo = mid + 1; else hi = mid; }
                z_merged[(size_t)r * (n + N) + m + lo] = v;
                merged_idx[(size_t)r * (n + N) + m + lo] = n + m;
            }
        }
    }
}

// ---- stage 2: z_final = sort([z_samples | near | far | z_vals[:, sel]]) and the main-pass points (:535-559) ----
__global__ void __launch_bounds__(256) sampler_finish_kernel(const float* __restrict__ z_samples, int Ns, const float* __restrict__ z_vals,
                                                             int n, const int32_t* __restrict__ sel, int Ne, float near, float far,
                                                             const float* __restrict__ cam_loc, const float* __restrict__ ray_dirs, int R,
                                                             float* __restrict__ z_out, float* __restrict__ points, const int32_t* __restrict__ flags,
                                                             int it) {
    if (flags && !(flags[it] != 0 && flags[it + 1] == 0)) return;      // runs behind the final sampling pass of the iteration the loop ends in
    __shared__ float smem[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    const bool active = r < R;
    const int M = Ns + 2 + Ne;
    float* c = smem + wave * 256;
    if (active)
        for (int q = lane; q < M; q += 64) {
            float v;
            if (q < Ns) v = z_samples[(size_t)r * Ns + q];
            else if (q == Ns) v = near;
            else if (q == Ns + 1) v = far;
            else v = z_vals[(size_t)r * n + sel[q - Ns - 2]];
            c[q] = v;
        }
    __syncthreads();
    if (!active) return;
    for (int q = lane; q < M; q += 64) {
        const float v = c[q], kv = okey(v);
        int rank = 0;
        for (int pth = 0; pth < M; ++pth) {
            const float w = okey(c[pth]);
            rank += (w < kv) || (w == kv && pth < q);
        }
        const size_t o = (size_t)r * M + rank;
        z_out[o] = v;
#pragma unroll
        for (int k = 0; k < 3; ++k) points[o * 3 + k] = cam_loc[3 * r + k] + v * ray_dirs[3 * r + k];
    }
}

}  // namespace

extern "C" {

int spf_sampler_uniform(const float* tlin, const float* t_rand, const float* cam_loc, const float* ray_dirs, int32_t R, int32_t n,
                        float near, float far, float* z, float* points, void* stream) {
    if (R < 0 || n < 1) return spf::fail(SPF_EINVAL, "spf_sampler_uniform: bad sizes");
    if (R == 0) return SPF_OK;
    if (!tlin || !cam_loc || !ray_dirs || !z || !points) return spf::fail(SPF_EINVAL, "spf_sampler_uniform: null pointer");
    uniform_kernel<<<spf::div_up((long long)R * n, 256), 256, 0, (hipStream_t)stream>>>(tlin, t_rand, cam_loc, ray_dirs, R, n, near, far, z, points);
    SPF_LAUNCH_CHECK("uniform_kernel");
    return SPF_OK;
}

int spf_sampler_iter(const float* z, const float* sdf, const float* beta_in, const float* beta0, int32_t R, int32_t n, float eps,
                     float bound_coef, int32_t beta_iters, int32_t more, float add_tiny, const float* u, int32_t u_per_ray, int32_t N,
                     float* samples, float* beta_out, float* z_merged, int32_t* merged_idx, int32_t* flags, int32_t it, void* stream) {
    if (R < 0 || n < 2 || n > 640 || N < 0 || (more && (N < 1 || N > 128))) return spf::fail(SPF_EINVAL, "spf_sampler_iter: need 2<=n<=640, N>=0 (1..128 when merging)");
    if (flags && (it < 0 || it > 30)) return spf::fail(SPF_EINVAL, "spf_sampler_iter: iteration index out of range");
    if (R == 0) return SPF_OK;
    if (!z || !sdf || !beta0 || !beta_out || (N > 0 && (!u || !samples)) || (more && (!z_merged || !merged_idx)))
        return spf::fail(SPF_EINVAL, "spf_sampler_iter: null pointer");
    const int blocks = spf::div_up(R, 4);
    hipStream_t s = (hipStream_t)stream;
    if (n <= 128)
        sampler_iter_kernel<2><<<blocks, 256, 0, s>>>(z, sdf, beta_in, beta0, R, n, eps, bound_coef, beta_iters, more, add_tiny, u, u_per_ray, N,
                                                       samples, beta_out, z_merged, merged_idx, flags, it, SamplerTail{});
    else
        sampler_iter_kernel<10><<<blocks, 256, 0, s>>>(z, sdf, beta_in, beta0, R, n, eps, bound_coef, beta_iters, more, add_tiny, u, u_per_ray, N,
                                                        samples, beta_out, z_merged, merged_idx, flags, it, SamplerTail{});
    SPF_LAUNCH_CHECK("sampler_iter_kernel");
    return SPF_OK;
}

int spf_sampler_train(const float* z, const float* pair_tmp, const int32_t* pair_off, const int32_t* slot_point, const float* beta0, int32_t R,
                      int32_t n, float eps, float bound_coef, int32_t beta_iters, const float* u, int32_t N, const int32_t* sel, int32_t Ne,
                      float near, float far, const float* cam_loc, const float* ray_dirs, const spf_grid* grid, int32_t SR, float* beta_out,
                      float* z_out, float* points, int32_t* slot_sample, uint8_t* ray_valid, void* stream) {
    if (R < 0 || n < 2 || n > 128 || N < 1 || N > 128 || Ne < 0 || N + 2 + Ne > 128 || SR < 1)
        return spf::fail(SPF_EINVAL, "spf_sampler_train: need 2 <= n <= 128, 1 <= N, N + 2 + Ne <= 128, SR >= 1");
    if (R == 0) return SPF_OK;
    if (!z || !pair_tmp || !pair_off || !slot_point || !beta0 || !u || (Ne > 0 && !sel) || !cam_loc || !ray_dirs || !grid || !beta_out || !z_out || !points ||
        !slot_sample || !ray_valid)
        return spf::fail(SPF_EINVAL, "spf_sampler_train: null pointer");
    if (grid->n_in == 0) return spf::fail(SPF_EINVAL, "spf_sampler_train: the grid holds no points");
    SamplerTail t{pair_tmp, pair_off, slot_point, sel, Ne, near, far, cam_loc, ray_dirs, z_out, points, spf::dev_view(grid), SR, slot_sample, ray_valid,
                  nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr};
    sampler_iter_kernel<2, MODE_TRAIN><<<spf::div_up(R, 4), 256, 0, (hipStream_t)stream>>>(z, nullptr, nullptr, beta0, R, n, eps, bound_coef, beta_iters, 0, 0.f, u, 1,
                                                                                           N, nullptr, beta_out, nullptr, nullptr, nullptr, 0, t);
    SPF_LAUNCH_CHECK("sampler_iter_kernel<fused>");
    return SPF_OK;
}

int spf_sampler_eval(const spf_sampler_eval_args* a, const spf_grid* grid, int32_t R, int32_t phase, void* stream) {
    if (!a) return spf::fail(SPF_EINVAL, "spf_sampler_eval: null arguments");
    const int n = a->n, n_new = n - a->n_prev;
    if (R < 0 || n < 2 || n > 640 || a->n_prev < 0 || n_new < 0 || phase < 0 || phase > 2 || a->it < 0 || a->it > 30)
        return spf::fail(SPF_EINVAL, "spf_sampler_eval: need 2 <= n <= 640, 0 <= n_prev <= n, phase in {0, 1, 2}, 0 <= it <= 30");
    if (R == 0) return SPF_OK;
    if (!a->z || !a->sdf_cur || !a->beta || !a->beta0 || !a->flags) return spf::fail(SPF_EINVAL, "spf_sampler_eval: null z / sdf_cur / beta / beta0 / flags");
    const bool gathers = phase != 1, finishes = phase != 0, merges = phase == 1;
    if (gathers && (!a->pair_tmp || !a->pair_off || !a->slot_point || (a->n_prev > 0 && (!a->sdf_prev || !a->merged_idx))))
        return spf::fail(SPF_EINVAL, "spf_sampler_eval: phases 0 / 2 need the pair scratch of the new samples and, with n_prev > 0, sdf_prev + merged_idx");
    if (merges && (!a->u_more || a->N_more < 1 || a->N_more > 128 || n + a->N_more > 640 || !a->z_merged || !a->merged_out || !a->points_new || !a->cam_loc || !a->ray_dirs))
        return spf::fail(SPF_EINVAL, "spf_sampler_eval: phase 1 needs u_more, 1 <= N_more <= 128, n + N_more <= 640, z_merged, merged_out, points_new, cam_loc, ray_dirs");
    if (finishes) {
        if (!a->u_fin || a->N_fin < 1 || a->Ne < 0 || a->N_fin + 2 + a->Ne > 128 || (a->Ne > 0 && !a->sel) || !a->cam_loc || !a->ray_dirs || !a->z_out || !a->points_out ||
            !a->slot_sample || !a->ray_valid || a->SR < 1 || !grid)
            return spf::fail(SPF_EINVAL, "spf_sampler_eval: phases 1 / 2 need u_fin, N_fin + 2 + Ne <= 128, sel, cam_loc, ray_dirs, z_out, points_out, slot_sample, ray_valid, SR, grid");
        if (grid->n_in == 0) return spf::fail(SPF_EINVAL, "spf_sampler_eval: the grid holds no points");
    }
    SamplerTail t{a->pair_tmp, a->pair_off, a->slot_point, a->sel, a->Ne, a->near, a->far, a->cam_loc, a->ray_dirs, a->z_out, a->points_out,
                  finishes ? spf::dev_view(grid) : GridDev{}, a->SR, a->slot_sample, a->ray_valid,
                  a->sdf_prev, a->n_prev > 0 ? a->merged_idx : nullptr, a->n_prev, gathers ? a->sdf_cur : nullptr, a->u_fin, a->N_fin, a->points_new};
    const int blocks = spf::div_up(R, 4);
    hipStream_t s = (hipStream_t)stream;
    const bool small = n <= 128;
#define SPF_EVAL_LAUNCH(MODE, sdf_in, beta_in, iters, more, u, N, zm, mi)                                                                                       \
    do {                                                                                                                                                        \
        if (small)                                                                                                                                              \
            sampler_iter_kernel<2, MODE><<<blocks, 256, 0, s>>>(a->z, sdf_in, beta_in, a->beta0, R, n, a->eps, a->bound_coef, iters, more, a->add_tiny, u, 0, N, \
                                                                nullptr, a->beta, zm, mi, a->flags, a->it, t);                                                   \
        else                                                                                                                                                    \
            sampler_iter_kernel<10, MODE><<<blocks, 256, 0, s>>>(a->z, sdf_in, beta_in, a->beta0, R, n, a->eps, a->bound_coef, iters, more, a->add_tiny, u, 0, N, \
                                                                 nullptr, a->beta, zm, mi, a->flags, a->it, t);                                                  \
    } while (0)
    // beta_in: the first iteration derives its starting beta from the sample spacing (ray_sampler.py:389-395), later ones continue from the last
    const float* beta_in = a->it == 0 ? nullptr : a->beta;
    if (phase == 0) SPF_EVAL_LAUNCH(MODE_EVAL_TEST, nullptr, beta_in, a->beta_iters, 0, nullptr, 0, nullptr, nullptr);
    else if (phase == 1) SPF_EVAL_LAUNCH(MODE_EVAL_STEP, a->sdf_cur, a->beta, 0, 1, a->u_more, a->N_more, a->z_merged, a->merged_out);
    else SPF_EVAL_LAUNCH(MODE_EVAL_LAST, nullptr, beta_in, a->beta_iters, 0, a->u_fin, a->N_fin, nullptr, nullptr);
#undef SPF_EVAL_LAUNCH
    SPF_LAUNCH_CHECK("sampler_iter_kernel<eval>");
    return SPF_OK;
}

int spf_sampler_finish(const float* z_samples, int32_t Ns, const float* z_vals, int32_t n, const int32_t* sel, int32_t Ne, float near,
                       float far, const float* cam_loc, const float* ray_dirs, int32_t R, float* z_out, float* points, const int32_t* flags,
                       int32_t it, void* stream) {
    if (R < 0 || Ns < 0 || Ne < 0 || Ns + 2 + Ne > 256) return spf::fail(SPF_EINVAL, "spf_sampler_finish: need Ns + 2 + Ne <= 256");
    if (R == 0) return SPF_OK;
    if (!z_samples || !z_vals || (Ne > 0 && !sel) || !cam_loc || !ray_dirs || !z_out || !points)
        return spf::fail(SPF_EINVAL, "spf_sampler_finish: null pointer");
    sampler_finish_kernel<<<spf::div_up(R, 4), 256, 0, (hipStream_t)stream>>>(z_samples, Ns, z_vals, n, sel, Ne, near, far, cam_loc, ray_dirs, R,
                                                                              z_out, points, flags, it);
    SPF_LAUNCH_CHECK("sampler_finish_kernel");
    return SPF_OK;
}

}  // extern "C"
