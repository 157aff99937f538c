// Scheduling experiment (not part of the library): a layer = [GEMM over 32 k-steps | epilogue that rewrites the activation tile].
//   A) two independent 4-wave workgroups per CU, each: GEMM, barrier, epilogue, barrier   (the tile engine's structure)
//   B) one 8-wave workgroup, two 4-wave groups one stage apart: while one group runs its GEMM the other runs its epilogue;
//      a workgroup barrier ends every stage
// hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf4p;
constexpr int LDA = 260;

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void gemm(const float* X, gf4p wp, int lane, f32x16 (&acc)[2][2]) {
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = X + i * LDA + 4 * h;
    const float* a1p = a0p + 32 * LDA;
    gf4p bp = wp + lane;
    f32x4 b0 = bp[0], b1 = bp[64], c0 = bp[128], c1 = bp[192];
    f32x4 a0 = *(const f32x4*)a0p, a1 = *(const f32x4*)a1p;
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
        f32x4 d0 = c0, d1 = c1, na0 = a0, na1 = a1;
        if (t + 2 < 32) { d0 = bp[(t + 2) * 128]; d1 = bp[(t + 2) * 128 + 64]; }
        if (t + 1 < 32) { na0 = *(const f32x4*)(a0p + 8 * (t + 1)); na1 = *(const f32x4*)(a1p + 8 * (t + 1)); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = c0; b1 = c1; c0 = d0; c1 = d1; a0 = na0; a1 = na1;
    }
}

template <int EPI = 3>   // bit 0: VALU work, bit 1: LDS writes
__device__ __forceinline__ void epilogue(float* X, f32x16 (&acc)[2][2], int wave, int lane) {
    const int c0 = wave * 64 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r];
                if (EPI & 1) {
                    v = v * 1e-3f + 0.01f;
                    v = v > 0.f ? v : v * 0.01f;
                }
                if (EPI & 2) X[(m * 32 + row_of(r, h)) * LDA + c0 + 32 * n] = v;
                acc[m][n][r] = (EPI & 2) ? 0.f : v * 0.5f;
            }
}

template <int EPI>
__global__ void __launch_bounds__(256, 2) kA(const float* w, float* out, int layers) {
    __shared__ __attribute__((aligned(16))) float X[64 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * LDA; e += 256) X[e] = 0.001f * (e & 127);
    __syncthreads();
    f32x16 acc[2][2] = {};
    for (int l = 0; l < layers; ++l) {
        gemm(X, (gf4p)w + ((l & 3) * 4 + wave) * (32 * 128), lane, acc);
        __syncthreads();
        if (EPI) epilogue<EPI>(X, acc, wave, lane);
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = X[tid] + acc[0][0][0];
}

__global__ void __launch_bounds__(512, 1) kB(const float* w, float* out, int layers) {
    __shared__ __attribute__((aligned(16))) float XX[2][64 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, grp = wave8 >> 2, wave = wave8 & 3;
    float* X = XX[grp];
    for (int e = tid & 255; e < 64 * LDA; e += 256) X[e] = 0.001f * (e & 127);
    __syncthreads();
    f32x16 acc[2][2] = {};
    // group 0: G E G E ...   group 1: (idle) G E G E ...: stage s -> group g runs GEMM when (s + g) is even
    for (int s = 0; s < 2 * layers + 1; ++s) {
        const int ls = s - grp;                // this group's own stage counter
        if (ls >= 0 && ls < 2 * layers) {
            if ((ls & 1) == 0) gemm(X, (gf4p)w + (((ls >> 1) & 3) * 4 + wave) * (32 * 128), lane, acc);
            else epilogue(X, acc, wave, lane);
        }
        __syncthreads();
    }
    out[blockIdx.x * 512 + tid] = X[tid & 255];
}

// D) structure A with the MFMA operands swapped (C^T = W X^T): a lane then holds 4 CONSECUTIVE output features of one row per
//    register quad, so the epilogue rewrites the tile with 16 ds_write_b128 per lane instead of 64 ds_write_b32
__device__ __forceinline__ void gemm_t(const float* X, gf4p wp, int lane, f32x16 (&acc)[2][2]) {
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = X + i * LDA + 4 * h;
    const float* a1p = a0p + 32 * LDA;
    gf4p bp = wp + lane;
    f32x4 b0 = bp[0], b1 = bp[64], c0 = bp[128], c1 = bp[192];
    f32x4 a0 = *(const f32x4*)a0p, a1 = *(const f32x4*)a1p;
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
        f32x4 d0 = c0, d1 = c1, na0 = a0, na1 = a1;
        if (t + 2 < 32) { d0 = bp[(t + 2) * 128]; d1 = bp[(t + 2) * 128 + 64]; }
        if (t + 1 < 32) { na0 = *(const f32x4*)(a0p + 8 * (t + 1)); na1 = *(const f32x4*)(a1p + 8 * (t + 1)); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {     // acc[m][n]: rows = features 32n.., columns = data rows 32m..
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a0[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a1[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a1[j], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = c0; b1 = c1; c0 = d0; c1 = d1; a0 = na0; a1 = na1;
    }
}

__device__ __forceinline__ void epilogue_t(float* X, f32x16 (&acc)[2][2], int wave, int lane) {
    const int row_l = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[m][n][4 * g + e] * 1e-3f + 0.01f;
                    v[e] = t > 0.f ? t : t * 0.01f;
                    acc[m][n][4 * g + e] = 0.f;
                }
                *(f32x4*)(X + (m * 32 + row_l) * LDA + wave * 64 + 32 * n + 8 * g + 4 * h) = v;
            }
}

__global__ void __launch_bounds__(256, 2) kD(const float* w, float* out, int layers) {
    __shared__ __attribute__((aligned(16))) float X[64 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * LDA; e += 256) X[e] = 0.001f * (e & 127);
    __syncthreads();
    f32x16 acc[2][2] = {};
    for (int l = 0; l < layers; ++l) {
        gemm_t(X, (gf4p)w + ((l & 3) * 4 + wave) * (32 * 128), lane, acc);
        __syncthreads();
        epilogue_t(X, acc, wave, lane);
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = X[tid] + acc[0][0][0];
}

// C) one 8-wave workgroup, both groups on the same layer at the same time; the layer's weight fragments are streamed ONCE per
//    workgroup into a 3-slot LDS ring by LDS-DMA (wave w of group 0 fetches the fragments both wave w's use) and read from there
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_dst_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__global__ void __launch_bounds__(512, 1) kC(const float* w, float* out, int layers) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 64 * LDA + 3 * 4 * 512];      // X[2] | ring[3 slots][4 waves][2 frags][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, grp = wave8 >> 2, wave = wave8 & 3;
    float* X = sm + grp * 64 * LDA;
    float* ring = sm + 2 * 64 * LDA;
    for (int e = tid & 255; e < 64 * LDA; e += 256) X[e] = 0.001f * (e & 127);
    __syncthreads();
    f32x16 acc[2][2] = {};
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = X + i * LDA + 4 * h;
    const float* a1p = a0p + 32 * LDA;
    for (int l = 0; l < layers; ++l) {
        const float* wl = w + (size_t)((l & 3) * 4 + wave) * (32 * 128 * 4);      // this wave pair's fragment stream of the layer
        auto issue = [&](int t) {
            if (grp == 0) {
                float* slot = ring + ((t % 3) * 4 + wave) * 512;
                glds16(wl + (size_t)(t * 128 + lane) * 4, slot);
                glds16(wl + (size_t)(t * 128 + 64 + lane) * 4, slot + 256);
            }
        };
        issue(0);
        issue(1);
        for (int t = 0; t < 32; ++t) {
            if (grp == 0) {
                if (t + 1 < 32) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                          // slot t landed; slot t-1 free again
            if (t + 2 < 32) issue(t + 2);
            const float* slot = ring + ((t % 3) * 4 + wave) * 512;
            const f32x4 b0 = *(const f32x4*)(slot + 4 * lane), b1 = *(const f32x4*)(slot + 256 + 4 * lane);
            const f32x4 a0 = *(const f32x4*)(a0p + 8 * t), a1 = *(const f32x4*)(a1p + 8 * t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();
        epilogue<3>(X, acc, wave, lane);
        __syncthreads();
    }
    out[blockIdx.x * 512 + tid] = X[tid & 255] + acc[0][0][0];
}

int main() {
    float *w, *out;
    const size_t wbytes = 16ull * 32 * 128 * 16;
    (void)hipMalloc(&w, wbytes); (void)hipMemset(w, 0, wbytes);
    (void)hipMalloc(&out, 4096 * 512 * 4);
    const int layers = 600;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
#define RUN_A(E, label)                                                                                                  \
        kA<E><<<512, 256>>>(w, out, layers);                                                                              \
        (void)hipEventRecord(e0); kA<E><<<512, 256>>>(w, out, layers); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                                           \
        printf("A two 4-wave workgroups/CU, %-28s: %.3f ms  %.1f TFLOP/s  %.2f us/layer\n", label, ms, 512.0 * layers * 64 * 256 * 256 * 2 / (ms * 1e-3) / 1e12, ms * 1e3 / layers);
        RUN_A(0, "no epilogue (barriers only)")
        RUN_A(1, "VALU only")
        RUN_A(2, "LDS writes only")
        RUN_A(3, "VALU + LDS writes")
        kB<<<256, 512>>>(w, out, layers);
        (void)hipEventRecord(e0); kB<<<256, 512>>>(w, out, layers); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("B one 8-wave workgroup, stage-shifted : %.3f ms  %.1f TFLOP/s\n", ms, 512.0 * layers * 64 * 256 * 256 * 2 / (ms * 1e-3) / 1e12);
        kD<<<512, 256>>>(w, out, layers);
        (void)hipEventRecord(e0); kD<<<512, 256>>>(w, out, layers); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("D structure A, transposed product + 16-byte tile rewrites : %.3f ms  %.1f TFLOP/s\n", ms, 512.0 * layers * 64 * 256 * 256 * 2 / (ms * 1e-3) / 1e12);
        kC<<<256, 512>>>(w, out, layers);
        (void)hipEventRecord(e0); kC<<<256, 512>>>(w, out, layers); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("C one 8-wave workgroup, weights via LDS-DMA ring : %.3f ms  %.1f TFLOP/s\n", ms, 512.0 * layers * 64 * 256 * 256 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
