// Measurement (not part of the library): the x3 GEMM loop of the tile engine (three bf16 pieces per operand, six piece products per
// output tile and k-step, weights streamed from L2, activations re-read from LDS planes, one 4-wave workgroup per CU, fp32 accumulate)
// written twice with the SAME operand traffic and the same 64 x 64 output per wave:
//   SHAPE 32: v_mfma_f32_32x32x16_bf16, 2 x 2 tiles, 24 MFMAs of 32 cycles per k16-step  (what mlp_tile_x3.h uses)
//   SHAPE 16: v_mfma_f32_16x16x32_bf16, 4 x 4 tiles, 96 MFMAs of 16 cycles per k32-step
// on random operands.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports that the 16x16x32 shape holds a higher clock under load at
// equal cycles per FLOP; this checks it for THIS loop (layers of 256 -> 256 with a barrier in between).  Prints wall TFLOP/s
// (fp32-equivalent: 2 x 64 x 256 x 256 per workgroup and layer), cycles per 32 k (1536 = matrix-pipe bound) and the held clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_shape_rate.hip -o mfma_shape_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef const __attribute__((address_space(1))) bf16x8* gptr;

constexpr int LDP = 264;                 // plane row stride (bf16), as X3_LDP
constexpr int PLANE = 64 * LDP;

template <int SHAPE>
__global__ void __launch_bounds__(256, 1) loop(const bf16x8* __restrict__ wfrag, const unsigned short* __restrict__ xinit, int layers,
                                               unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 X[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 3 * PLANE; e += 256) reinterpret_cast<unsigned short*>(X)[e] = xinit[e];
    __syncthreads();
    // per wave and layer: 8 k32-steps x 12 fragments x 64 lanes (98 KB), the same bytes in both shapes
    gptr wp = (gptr)wfrag + (size_t)wave * (8 * 12 * 64) + lane;
    float keep = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (SHAPE == 32) {
        const int j = lane & 31, kg = lane >> 5;
        const __bf16* xp = X + j * LDP + 8 * kg;
        f32x16 acc[2][2];
        for (int l = 0; l < layers; ++l) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
            bf16x8 w[2][2][3], x[2][2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    w[0][m][p] = wp[(m * 3 + p) * 64];
                    x[0][m][p] = *reinterpret_cast<const bf16x8*>(xp + p * PLANE + 32 * m * LDP);
                }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int c = t & 1, nx = c ^ 1;
                if (t + 1 < 16) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            w[nx][m][p] = wp[((t + 1) * 6 + m * 3 + p) * 64];
                            x[nx][m][p] = *reinterpret_cast<const bf16x8*>(xp + p * PLANE + 32 * m * LDP + 16 * (t + 1));
                        }
                }
#define P32(PW, PX)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 2; ++n)             \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[c][m][PW], x[c][n][PX], acc[m][n], 0, 0, 0);
                P32(2, 0) P32(0, 2) P32(1, 1) P32(1, 0) P32(0, 1) P32(0, 0)
#undef P32
            }
            keep += (acc[0][0][0] + acc[0][1][1]) + (acc[1][0][2] + acc[1][1][3]);
            __syncthreads();
        }
    } else {
        const int j = lane & 15, kg = lane >> 4;
        const __bf16* xp = X + j * LDP + 8 * kg;
        f32x4 acc[4][4];
        for (int l = 0; l < layers; ++l) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 w[2][4][3], x[2][4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    w[0][a][p] = wp[(a * 3 + p) * 64];
                    x[0][a][p] = *reinterpret_cast<const bf16x8*>(xp + p * PLANE + 16 * a * LDP);
                }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = t & 1, nx = c ^ 1;
                if (t + 1 < 8) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            w[nx][a][p] = wp[((t + 1) * 12 + a * 3 + p) * 64];
                            x[nx][a][p] = *reinterpret_cast<const bf16x8*>(xp + p * PLANE + 16 * a * LDP + 32 * (t + 1));
                        }
                }
#define P16(PW, PX)                                                                                           \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b)             \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][a][PW], x[c][b][PX], acc[a][b], 0, 0, 0);
                P16(2, 0) P16(0, 2) P16(1, 1) P16(1, 0) P16(0, 1) P16(0, 0)
#undef P16
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) keep += acc[a][b][(a + b) & 3];          // every accumulator chain stays live
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (keep == 123.456f) sink[0] = keep;
}

static unsigned short bf16_bits(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    const size_t nfrag = (size_t)4 * 8 * 12 * 64;
    std::vector<unsigned short> hw(nfrag * 8), hx(3 * PLANE);
    srand(1);
    for (auto& v : hw) v = zeros ? 0 : bf16_bits(0.1f * ((float)rand() / RAND_MAX * 2.f - 1.f));
    for (auto& v : hx) v = zeros ? 0 : bf16_bits(0.1f * ((float)rand() / RAND_MAX * 2.f - 1.f));
    bf16x8* dW; unsigned short* dX; unsigned long long* dC; float* dS;
    (void)hipMalloc(&dW, nfrag * 16); (void)hipMemcpy(dW, hw.data(), nfrag * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&dX, hx.size() * 2); (void)hipMemcpy(dX, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&dC, 256 * 8); (void)hipMalloc(&dS, 4);
    const int layers = 20000;       // ~0.2 s per launch: long enough for the clock to settle
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        for (int shape : {32, 16, 16, 32}) {
            float ms; std::vector<unsigned long long> c(256);
            for (int i = 0; i < 3; ++i) {          // the third launch is reported
                (void)hipEventRecord(e0);
                if (shape == 32) loop<32><<<256, 256>>>(dW, dX, layers, dC, dS); else loop<16><<<256, 256>>>(dW, dX, layers, dC, dS);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(c.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
            double cm = 0; for (auto v : c) cm += (double)v; cm /= 256;
            printf("%s %s: %.1f ms, %.1f fp32-equivalent TFLOP/s, %.0f cycles per 32 k (1536 = pipe bound), %.3f GHz\n", zeros ? "zeros " : "random",
                   shape == 32 ? "32x32x16" : "16x16x32", ms, 256.0 * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12, cm / layers / 8.0, cm / (ms * 1e6));
        }
    }
    return 0;
}
