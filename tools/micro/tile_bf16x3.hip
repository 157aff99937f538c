// Prototype (not part of the library): the MLP tile engine's layer loop with fp32-exact products from three bf16 pieces per
// operand — the follow-up named in DESIGN.md §8.  One 4-wave workgroup per CU, tile = 64 rows, layer = 256 -> 256:
//   D[feature][row] += W[feature][k] X[k][row]  on v_mfma_f32_32x32x16_bf16 (transposed product: a lane ends up with 4
//   consecutive features of ONE row per register quad, so the epilogue writes packed 8-byte bf16 quads);
//   activations live in LDS as three bf16 planes [piece][row][k] (16-byte operand reads), weights stream from L2 as packed
//   piece fragments [layer][wave][k16][m][piece][lane] x 16 B; epilogue = + bias, LeakyReLU, split into three pieces, write.
// Reports fp32-equivalent TFLOP/s (2*64*256*256 per layer-tile) and checks two layers against a float64 host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tile_bf16x3.hip -o tile_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef const __attribute__((address_space(1))) bf16x8* gfrag;

constexpr int ROWS = 64, K = 256, LDP = 264;            // LDP: plane row stride in bf16 (528 B: 16-B aligned, off the 256-B period)
constexpr int PLANE = ROWS * LDP;                        // bf16 elements per plane

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)x;
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    c = (__bf16)(r1 - (float)b);
}

// weights: [layer][wave 4][t 16][m 2][piece 3][lane 64] bf16x8;  lane (i = l & 31, kg = l >> 5) <-> W[64 w + 32 m + i][16 t + 8 kg + e]
__global__ void __launch_bounds__(256, 1) k(const bf16x8* __restrict__ wfrag, const float* __restrict__ bias, const float* __restrict__ x0, float* out,
                                            int layers, int nlayer_w) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // initial activations: x0 [64][256] fp32 -> planes
    for (int e = tid; e < ROWS * K; e += 256) {
        __bf16 a, b, c;
        split3(x0[(size_t)blockIdx.x % 1 * 0 + e], a, b, c);
        const int row = e >> 8, col = e & 255;
        X[0 * PLANE + row * LDP + col] = a;
        X[1 * PLANE + row * LDP + col] = b;
        X[2 * PLANE + row * LDP + col] = c;
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        gfrag wp = (gfrag)wfrag + ((size_t)(l % nlayer_w) * 4 + wave) * (16 * 2 * 3 * 64) + lane;
        bf16x8 wa[2][3], wn[2][3], wnn[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) { wa[m][p] = wp[(m * 3 + p) * 64]; wn[m][p] = wp[(6 + m * 3 + p) * 64]; }
        bf16x8 xb[2][3], xn[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) xb[n][p] = *(const bf16x8*)(X + p * PLANE + (32 * n + j) * LDP + 8 * kg);
#pragma unroll 2
        for (int t = 0; t < 16; ++t) {
            if (t + 2 < 16) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) wnn[m][p] = wp[((t + 2) * 6 + m * 3 + p) * 64];
            }
            if (t + 1 < 16) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int p = 0; p < 3; ++p) xn[n][p] = *(const bf16x8*)(X + p * PLANE + (32 * n + j) * LDP + 16 * (t + 1) + 8 * kg);
            }
            __builtin_amdgcn_sched_barrier(0);
#define MMA(PW, PX)                                                                                                  \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PW], xb[0][PX], acc[0][0], 0, 0, 0);                    \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][PW], xb[1][PX], acc[0][1], 0, 0, 0);                    \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PW], xb[0][PX], acc[1][0], 0, 0, 0);                    \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][PW], xb[1][PX], acc[1][1], 0, 0, 0);
            MMA(2, 0) MMA(0, 2) MMA(1, 1) MMA(1, 0) MMA(0, 1) MMA(0, 0)
#undef MMA
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) { wa[m][p] = wn[m][p]; wn[m][p] = wnn[m][p]; }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[n][p] = xn[n][p];
        }
        __syncthreads();
        // epilogue: acc[m][n][4g + e] = feature 64w + 32m + 8g + 4kg + e of row 32n + j
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
                const f32x4 bv = *(const f32x4*)(bias + (size_t)(l % nlayer_w) * 256 + f0);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bf16x4 pa, pb, pc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[m][n][4 * g + e] + bv[e];
                        v = v > 0.f ? v : v * 0.01f;
                        __bf16 a, b, c;
                        split3(v, a, b, c);
                        pa[e] = a; pb[e] = b; pc[e] = c;
                    }
                    __bf16* dst = X + (32 * n + j) * LDP + f0;
                    *(bf16x4*)(dst) = pa;
                    *(bf16x4*)(dst + PLANE) = pb;
                    *(bf16x4*)(dst + 2 * PLANE) = pc;
                }
            }
        __syncthreads();
    }
    if (blockIdx.x == 0)
        for (int e = tid; e < ROWS * K; e += 256) {
            const int row = e >> 8, col = e & 255;
            out[e] = (float)X[row * LDP + col] + (float)X[PLANE + row * LDP + col] + (float)X[2 * PLANE + row * LDP + col];
        }
}

// 8-wave variant: wave w owns features [32w, 32w+32) x 64 rows; weights [layer][wave 8][t 16][piece 3][lane 64]
__global__ void __launch_bounds__(512, 1) k8(const bf16x8* __restrict__ wfrag, const float* __restrict__ bias, const float* __restrict__ x0, float* out,
                                             int layers, int nlayer_w) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < ROWS * K; e += 512) {
        __bf16 a, b, c;
        split3(x0[e], a, b, c);
        const int row = e >> 8, col = e & 255;
        X[0 * PLANE + row * LDP + col] = a;
        X[1 * PLANE + row * LDP + col] = b;
        X[2 * PLANE + row * LDP + col] = c;
    }
    __syncthreads();
    f32x16 acc[2];
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        // same fragment image as k: [layer][wave4][t][m][piece][lane]; wave8 = 2 * wave4 + m
        gfrag wp = (gfrag)wfrag + (((size_t)(l % nlayer_w) * 4 + (wave >> 1)) * (16 * 2 * 3) + (wave & 1) * 3) * 64 + lane;
        bf16x8 wa[3], wn[3], wnn[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { wa[p] = wp[p * 64]; wn[p] = wp[(6 + p) * 64]; }
#pragma unroll 2
        for (int t = 0; t < 16; ++t) {
            if (t + 2 < 16) {
#pragma unroll
                for (int p = 0; p < 3; ++p) wnn[p] = wp[((t + 2) * 6 + p) * 64];
            }
            bf16x8 xb[2][3];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 3; ++p) xb[n][p] = *(const bf16x8*)(X + p * PLANE + (32 * n + j) * LDP + 16 * t + 8 * kg);
            __builtin_amdgcn_sched_barrier(0);
#define MMA(PW, PX)                                                                                        \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[PW], xb[0][PX], acc[0], 0, 0, 0);                   \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[PW], xb[1][PX], acc[1], 0, 0, 0);
            MMA(2, 0) MMA(0, 2) MMA(1, 1) MMA(1, 0) MMA(0, 1) MMA(0, 0)
#undef MMA
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) { wa[p] = wn[p]; wn[p] = wnn[p]; }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * wave + 8 * g + 4 * kg;
            const f32x4 bv = *(const f32x4*)(bias + (size_t)(l % nlayer_w) * 256 + f0);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bf16x4 pa, pb, pc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[n][4 * g + e] + bv[e];
                    v = v > 0.f ? v : v * 0.01f;
                    __bf16 a, b, c;
                    split3(v, a, b, c);
                    pa[e] = a; pb[e] = b; pc[e] = c;
                }
                __bf16* dst = X + (32 * n + j) * LDP + f0;
                *(bf16x4*)(dst) = pa;
                *(bf16x4*)(dst + PLANE) = pb;
                *(bf16x4*)(dst + 2 * PLANE) = pc;
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0)
        for (int e = tid; e < ROWS * K; e += 512) {
            const int row = e >> 8, col = e & 255;
            out[e] = (float)X[row * LDP + col] + (float)X[PLANE + row * LDP + col] + (float)X[2 * PLANE + row * LDP + col];
        }
}

static float bf16_round(float x) { __bf16 b = (__bf16)x; return (float)b; }

int main() {
    const int NLW = 4;
    std::vector<float> W((size_t)NLW * 256 * 256), B((size_t)NLW * 256), X0(ROWS * K);
    srand(1);
    for (auto& v : W) v = ((rand() % 20001) - 10000) / 10000.0f * 0.11f;
    for (auto& v : B) v = ((rand() % 2001) - 1000) / 1000.0f * 0.1f;
    for (auto& v : X0) v = ((rand() % 20001) - 10000) / 10000.0f;
    // pack weights into piece fragments
    std::vector<__bf16> frag((size_t)NLW * 4 * 16 * 2 * 3 * 64 * 8);
    for (int l = 0; l < NLW; ++l)
        for (int w = 0; w < 4; ++w)
            for (int t = 0; t < 16; ++t)
                for (int m = 0; m < 2; ++m)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const int i = ln & 31, kg = ln >> 5;
                            const float x = W[((size_t)l * 256 + 64 * w + 32 * m + i) * 256 + 16 * t + 8 * kg + e];
                            const float a = bf16_round(x), r1 = x - a, b = bf16_round(r1), c = bf16_round(r1 - b);
                            const float pc[3] = {a, b, c};
                            for (int p = 0; p < 3; ++p)
                                frag[((((((size_t)l * 4 + w) * 16 + t) * 2 + m) * 3 + p) * 64 + ln) * 8 + e] = (__bf16)pc[p];
                        }
    bf16x8* dW; float *dB, *dX, *dO;
    (void)hipMalloc(&dW, frag.size() * 2); (void)hipMemcpy(dW, frag.data(), frag.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&dB, B.size() * 4); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dX, X0.size() * 4); (void)hipMemcpy(dX, X0.data(), X0.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dO, X0.size() * 4);
    // correctness: two layers against float64
    k<<<1, 256>>>(dW, dB, dX, dO, 2, NLW);
    std::vector<float> got(ROWS * K);
    (void)hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost);
    std::vector<double> cur(X0.begin(), X0.end()), nxt(ROWS * K);
    for (int l = 0; l < 2; ++l) {
        for (int r = 0; r < ROWS; ++r)
            for (int f = 0; f < 256; ++f) {
                double s = B[l * 256 + f];
                for (int kk = 0; kk < 256; ++kk) s += (double)W[((size_t)l * 256 + f) * 256 + kk] * cur[r * 256 + kk];
                nxt[r * 256 + f] = (double)(float)(s > 0 ? s : 0.01 * s);       // activations are stored in fp32 precision (3 x 8 bits)
            }
        cur = nxt;
    }
    double emax = 0, vmax = 0;
    for (int e = 0; e < ROWS * K; ++e) { emax = fmax(emax, fabs(got[e] - cur[e])); vmax = fmax(vmax, fabs(cur[e])); }
    printf("two layers vs float64: max |err| %.3e on max |value| %.3e  (rel %.2e)\n", emax, vmax, emax / vmax);
    // speed
    const int layers = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 256>>>(dW, dB, dX, dO, layers, NLW);
    (void)hipEventRecord(e0); k<<<256, 256>>>(dW, dB, dX, dO, layers, NLW); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("one 4-wave workgroup per CU: %.3f ms for %d layers -> %.1f fp32-equivalent TFLOP/s (%.2f us per layer-tile)\n", ms, layers,
           256.0 * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12, ms * 1e3 / layers);
    k8<<<1, 512>>>(dW, dB, dX, dO, 2, NLW);
    (void)hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost);
    emax = 0;
    for (int e = 0; e < ROWS * K; ++e) emax = fmax(emax, fabs(got[e] - cur[e]));
    k8<<<256, 512>>>(dW, dB, dX, dO, layers, NLW);
    (void)hipEventRecord(e0); k8<<<256, 512>>>(dW, dB, dX, dO, layers, NLW); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("one 8-wave workgroup per CU: %.3f ms for %d layers -> %.1f fp32-equivalent TFLOP/s (%.2f us per layer-tile), max |err| %.2e\n", ms, layers,
           256.0 * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12, ms * 1e3 / layers, emax);
    return 0;
}
