// Issue-rate experiment: v_mfma_f32_32x32x16_bf16 back to back, one wave per SIMD, with 4 or 16 accumulator tiles (the latter
// lives in AGPRs).   hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ void __launch_bounds__(256, 1) k(const float* w, float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC] = {};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)w[threadIdx.x + e]; b[e] = (__bf16)w[threadIdx.x + 8 + e]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(const float* w, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC><<<256, 256>>>(w, out, 10, cyc);
    (void)hipEventRecord(e0); k<NACC><<<256, 256>>>(w, out, iters, cyc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 6 * NACC;
    printf("%2d accumulator tiles: %.1f ns/MFMA/wave, %.1f ticks/MFMA, %.0f TFLOP/s (bf16)\n", NACC, ms * 1e6 / n, (double)c / n,
           256.0 * 4 * n * 32768 / (ms * 1e-3) / 1e12);
}

int main() {
    float *w, *out; unsigned long long* cyc;
    (void)hipMalloc(&w, 4096); (void)hipMemset(w, 0, 4096); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    run<4>(w, out, cyc);
    run<16>(w, out, cyc);
    return 0;
}
