// Prototype / measurement (not part of the library): how fast does the library's x3 GEMM loop (mlp_tile_x3.h: gemm_x3<16>) run when
// nothing else is in the kernel?  One 4-wave workgroup per CU, planes filled once, `layers` back-to-back 256 -> 256 GEMMs with
//   (a) only a barrier between them,  (b) the minimal epilogue (accumulators -> planes, packed split),  (c) (b) + bias / LeakyReLU / sign words.
// Reports fp32-equivalent TFLOP/s and cycles per k-step (s_memtime) so that the clock under this load can be read off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../spurfies_amd/csrc -I../../include x3_loop_rate.hip -o x3_loop_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "mlp_tile_x3.h"
using namespace spf;

// ablations of the loop's operand traffic (wrong results, timing only): VAR 1 = every k-step re-reads the k-step-0 weight fragments
// (same request count, one 6 KB line set per wave: L1-resident instead of streamed from L2), VAR 2 = LDS operand of k-step 0 every step
template <int VAR>
__device__ __forceinline__ WFrag3 gemm_var(const __bf16* X, gx3 wp, int lane, f32x16 (&acc)[2][2], const WFrag3& first) {
    const int j = lane & 31, kg = lane >> 5;
    const __bf16* xp = X + j * X3_LDP + 8 * kg;
    X3Regs r;
    WFrag3 nxt = first;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) { r.w[0][m][p] = first.w[m][p]; r.w[1][m][p] = first.w1[m][p]; }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int p = 0; p < 3; ++p) r.x[0][n][p] = *reinterpret_cast<const bf16x8*>(xp + p * X3_PLANE + 32 * n * X3_LDP);
    gx3 wq = VAR == 1 ? wp - 12 * 64 : wp;           // (t + 2) * 6 * 64 with t = 0 lands on k-step 0 ... keep the address arithmetic, pin t
    x3_step<0, true, true, 0, false, X3_LDP, true>(VAR == 2 ? xp - 16 : xp, wq, 0, acc, r, nxt, wp);
    x3_step<1, true, true, 0>(VAR == 2 ? xp - 32 : xp, VAR == 1 ? wq - 6 * 64 : wq, 1, acc, r, nxt, wp);
    x3_step<2, true, true, 0>(VAR == 2 ? xp - 48 : xp, VAR == 1 ? wq - 12 * 64 : wq, 2, acc, r, nxt, wp);
#pragma unroll 1
    for (int t = 3; t + 3 <= 14; t += 3) {
        const int tw = VAR == 1 ? 0 : t, tx = VAR == 2 ? 0 : t;
        x3_step<0, true, true, 0>(xp + 16 * (tx - t), wp + (tw - t) * 6 * 64 - (VAR == 1 ? 12 * 64 : 0), t, acc, r, nxt, wp);
        x3_step<1, true, true, 0>(xp + 16 * (tx - t) - (VAR == 2 ? 16 : 0), wp + (tw - t) * 6 * 64 - (VAR == 1 ? 18 * 64 : 0), t + 1, acc, r, nxt, wp);
        x3_step<2, true, true, 0>(xp + 16 * (tx - t) - (VAR == 2 ? 32 : 0), wp + (tw - t) * 6 * 64 - (VAR == 1 ? 24 * 64 : 0), t + 2, acc, r, nxt, wp);
    }
    x3_step<0, true, true, 0>(VAR == 2 ? xp - 16 * 13 : xp, VAR == 1 ? wp - 14 * 6 * 64 : wp, 12, acc, r, nxt, wp);
    x3_step<1, true, true, 0>(VAR == 2 ? xp - 16 * 14 : xp, VAR == 1 ? wp - 15 * 6 * 64 : wp, 13, acc, r, nxt, wp);
    x3_step<2, false, true, 1>(VAR == 2 ? xp - 16 * 15 : xp, wp, 14, acc, r, nxt, wp);
    x3_step<0, false, false, 2>(xp, wp, 15, acc, r, nxt, wp);
    return nxt;
}

template <int MODE, int VAR = 0>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* __restrict__ wfrag, const float* __restrict__ bias, int layers, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * X3_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 3 * X3_PLANE; e += 256) X[e] = (__bf16)(0.001f * (float)((e * 37) % 101 - 50));
    __syncthreads();
    gx3 wp = (gx3)wfrag + wave * (16 * 2 * 3 * 64) + lane;
    WFrag3 nf = load_wfrag3(wp);
    f32x16 acc[2][2];
    uint32_t bits[2] = {0u, 0u};
    float keep = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < layers; ++l) {
        if (VAR == 0) nf = gemm_x3<16>(X, wp, lane, acc, nf, wp);
        else nf = gemm_var<VAR>(X, wp, lane, acc, nf);
        keep += (acc[0][0][0] + acc[0][1][0]) + (acc[1][0][0] + acc[1][1][0]);      // every accumulator chain stays live
        lds_barrier();
        if (MODE >= 1) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x4 out;
                        if (MODE == 2) {
                            f32x4 h, hs;
                            bias_scale4(acc[m][n], g, *reinterpret_cast<const f32x4*>(bias + f0), h, hs);
#pragma unroll
                            for (int e = 0; e < 4; ++e) out[e] = lrelu_push(h[e], hs[e], bits[n]);
                        } else {
                            out = f32x4{acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]} * 1e-3f;
                        }
                        store_quad_x3(X, 32 * n + j, f0, out);
                    }
                }
            lds_barrier();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (keep == 123.456f || bits[0] == 0x12345u) cyc[1000] = 1;     // keep the results alive
}

int main(int argc, char** argv) {
    const size_t nfrag = (size_t)4 * 16 * 2 * 3 * 64;
    std::vector<unsigned short> h(nfrag * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 7) % 64);     // small bf16 values
    bf16x8* dW; float* dB; unsigned long long* dC;
    (void)hipMalloc(&dW, nfrag * 16); (void)hipMemcpy(dW, h.data(), nfrag * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&dB, 1024); (void)hipMemset(dB, 0, 1024);
    (void)hipMalloc(&dC, 1001 * 8);
    const int layers = 600;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms; unsigned long long c0;
    if (argc >= 3) {
        // sustained mode for tools/power_probe.py: `x3_loop_rate <0|1|2> <seconds>` runs ONE loop body back to back for that long (after 1 s of
        // untimed load) and prints the averages, so that a power sampler beside it sees a steady state
        const int mode = atoi(argv[1]);
        const double secs = atof(argv[2]);
        auto launch = [&]() {
            if (mode == 0) k<0><<<256, 256>>>(dW, dB, layers, dC);
            else if (mode == 1) k<1><<<256, 256>>>(dW, dB, layers, dC);
            else k<2><<<256, 256>>>(dW, dB, layers, dC);
        };
        double tot_ms = 0, tot_cyc = 0; int n = 0;
        for (int phase = 0; phase < 2; ++phase) {
            const double want = phase == 0 ? 1000.0 : secs * 1000.0;
            double got = 0;
            while (got < want) {
                (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&c0, dC, 8, hipMemcpyDeviceToHost);
                got += ms;
                if (phase == 1) { tot_ms += ms; tot_cyc += (double)c0; ++n; }
            }
        }
        printf("{\"mode\": %d, \"launches\": %d, \"ms_per_launch\": %.4f, \"tflops_fp32_equiv\": %.2f, \"cycles_per_kstep\": %.1f, \"ghz\": %.4f, \"mfma_duty\": %.4f}\n",
               mode, n, tot_ms / n, 256.0 * layers * 2.0 * 64 * 256 * 256 * n / (tot_ms * 1e-3) / 1e12, tot_cyc / n / layers / 16.0, tot_cyc / (tot_ms * 1e6),
               768.0 / (tot_cyc / n / layers / 16.0));
        return 0;
    }
#define RUN(M, label)                                                                                                              \
    k<M><<<256, 256>>>(dW, dB, layers, dC);                                                                                        \
    (void)hipEventRecord(e0); k<M><<<256, 256>>>(dW, dB, layers, dC); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);     \
    (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&c0, dC, 8, hipMemcpyDeviceToHost);                                    \
    printf(label ": %.3f ms, %.1f fp32-equivalent TFLOP/s, %.0f cycles per k-step (768 = matrix-pipe bound), %.2f GHz\n", ms,     \
           256.0 * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12, (double)c0 / layers / 16.0, (double)c0 / (ms * 1e6));
    RUN(0, "GEMM + barrier                      ")
    RUN(1, "GEMM + minimal epilogue (split only)")
    RUN(2, "GEMM + bias / LeakyReLU / sign words")
#undef RUN
#define RUN(M, V, label)                                                                                                           \
    k<M, V><<<256, 256>>>(dW, dB, layers, dC);                                                                                     \
    (void)hipEventRecord(e0); k<M, V><<<256, 256>>>(dW, dB, layers, dC); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);  \
    (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(&c0, dC, 8, hipMemcpyDeviceToHost);                                    \
    printf(label ": %.3f ms, %.1f fp32-equivalent TFLOP/s, %.0f cycles per k-step, %.2f GHz\n", ms,                                \
           256.0 * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12, (double)c0 / layers / 16.0, (double)c0 / (ms * 1e6));
    RUN(2, 1, "  same, weights not streamed (k-step-0 fragments every step)")
    RUN(2, 2, "  same, LDS operand of k-step 0 every step                  ")
    return 0;
}
