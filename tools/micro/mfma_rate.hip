// Issue-rate experiment (not part of the library): cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD vs two, with and
// without the LDS / global operand traffic of the tile engine's k-step.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf4p;

template <int MODE>   // 0: registers only, 1: + LDS operand reads, 2: + global weight reads, 3: both
__global__ void __launch_bounds__(256, 2) k(const float* w, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float X[64 * 260];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 64 * 260; e += 256) X[e] = 0.001f * (e & 127);
    __syncthreads();
    f32x16 acc[2][2] = {};
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = X + i * 260 + 4 * h;
    const float* a1p = a0p + 32 * 260;
    gf4p bp = (gf4p)w + wave * (32 * 128) + lane;
    f32x4 b0 = bp[0], b1 = bp[64];
    f32x4 a0 = *(const f32x4*)a0p, a1 = *(const f32x4*)a1p;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int t = 0; t < 32; ++t) {
            f32x4 nb0 = b0, nb1 = b1, na0 = a0, na1 = a1;
            if (MODE & 2) { nb0 = bp[((t + 1) & 31) * 128]; nb1 = bp[((t + 1) & 31) * 128 + 64]; }
            if (MODE & 1) { na0 = *(const f32x4*)(a0p + 8 * ((t + 1) & 31)); na1 = *(const f32x4*)(a1p + 8 * ((t + 1) & 31)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
            b0 = nb0; b1 = nb1; a0 = na0; a1 = na1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, const float* w, float* out, unsigned long long* cyc) {
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(w, out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(w, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 32 * 16;
    printf("%-28s blocks=%4d  %.3f ms  %.1f ns/MFMA/wave  %.1f s_memtime ticks/MFMA  -> %.1f TFLOP/s\n", name, blocks, ms, ms * 1e6 / mf, (double)c / mf,
           (double)blocks * 4 * mf * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 4 * 32 * 128 * 16 + 4096); hipMemset(w, 0, 4 * 32 * 128 * 16 + 4096);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 8);
    for (int blocks : {256, 512}) {      // one / two workgroups per CU (66.6 KB LDS each)
        run<0>("registers only", blocks, w, out, cyc);
        run<1>("+ LDS operand reads", blocks, w, out, cyc);
        run<2>("+ global weight reads", blocks, w, out, cyc);
        run<3>("+ both", blocks, w, out, cyc);
    }
    return 0;
}
