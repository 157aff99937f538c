// Prototype (not part of the library): the bf16-piece MLP tile engine with the activations kept in LDS as FP32 and split into the
// three bf16 pieces in registers, per k-step, by the wave that consumes them.  A tile then costs 66.5 KB of LDS instead of 101 KB,
// TWO workgroups fit a CU, and one workgroup's epilogue / split arithmetic overlaps the other's MFMAs (two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tile_x3f.hip -o tile_x3f
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) bf16x8* gfrag;

constexpr int ROWS = 64, K = 256, LDF = 260;

__device__ __forceinline__ uint32_t cvt_pk_bf16(f32x2 a) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)); }
__device__ __forceinline__ f32x2 pk_bf16_to_f32(uint32_t c) { return f32x2{__uint_as_float(c << 16), __uint_as_float(c & 0xffff0000u)}; }
__device__ __forceinline__ void split3_pair(f32x2 a, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
    c1 = cvt_pk_bf16(a);
    const f32x2 r1 = a - pk_bf16_to_f32(c1);
    c2 = cvt_pk_bf16(r1);
    const f32x2 r2 = r1 - pk_bf16_to_f32(c2);
    c3 = cvt_pk_bf16(r2);
}
// 8 consecutive k of one row (two float4) -> the three bf16x8 MFMA operands
__device__ __forceinline__ void split8(f32x4 lo, f32x4 hi, bf16x8 (&x)[3]) {
    uint32_t c[3][4];
    split3_pair(f32x2{lo[0], lo[1]}, c[0][0], c[1][0], c[2][0]);
    split3_pair(f32x2{lo[2], lo[3]}, c[0][1], c[1][1], c[2][1]);
    split3_pair(f32x2{hi[0], hi[1]}, c[0][2], c[1][2], c[2][2]);
    split3_pair(f32x2{hi[2], hi[3]}, c[0][3], c[1][3], c[2][3]);
#pragma unroll
    for (int p = 0; p < 3; ++p) x[p] = __builtin_bit_cast(bf16x8, (u32x4{c[p][0], c[p][1], c[p][2], c[p][3]}));
}

template <int WGS, bool IL>
__global__ void __launch_bounds__(256, WGS) kf(const bf16x8* __restrict__ wfrag, const float* __restrict__ bias, const float* __restrict__ x0, float* out,
                                              int layers, int nlayer_w) {
    __shared__ __attribute__((aligned(16))) float X[ROWS * LDF];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < ROWS * K; e += 256) X[(e >> 8) * LDF + (e & 255)] = x0[e];
    __syncthreads();
    const float* xp = X + j * LDF + 8 * kg;
    f32x16 acc[2][2];
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        gfrag wp = (gfrag)wfrag + ((size_t)(l % nlayer_w) * 4 + wave) * (16 * 2 * 3 * 64) + lane;
        bf16x8 w[3][2][3];          // three phases (t mod 3)
        f32x4 raw[2][2][2];         // [phase t mod 2][n][half]: fp32 operands, requested one k-step before they are split
        bf16x8 x[2][2][3];          // [phase][n][piece]
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) { w[0][m][p] = wp[(m * 3 + p) * 64]; w[1][m][p] = wp[(6 + m * 3 + p) * 64]; }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x4 a = *(const f32x4*)(xp + 32 * n * LDF), b = *(const f32x4*)(xp + 32 * n * LDF + 4);
            split8(a, b, x[0][n]);
            raw[1][n][0] = *(const f32x4*)(xp + 32 * n * LDF + 16);
            raw[1][n][1] = *(const f32x4*)(xp + 32 * n * LDF + 20);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int R = t % 3, R2 = (t + 2) % 3, P = t & 1, P1 = (t + 1) & 1;
            if (t + 2 < 16) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) w[R2][m][p] = wp[((t + 2) * 6 + m * 3 + p) * 64];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    raw[P][n][0] = *(const f32x4*)(xp + 32 * n * LDF + 16 * (t + 2));
                    raw[P][n][1] = *(const f32x4*)(xp + 32 * n * LDF + 16 * (t + 2) + 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 16) {        // split the operands of the next k-step (requested during the previous one)
#pragma unroll
                for (int n = 0; n < 2; ++n) split8(raw[P1][n][0], raw[P1][n][1], x[P1][n]);
            }
#define MMA(PW, PX)                                                                                                  \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[R][0][PW], x[P][0][PX], acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[R][0][PW], x[P][1][PX], acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[R][1][PW], x[P][0][PX], acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[R][1][PW], x[P][1][PX], acc[1][1], 0, 0, 0);
            MMA(2, 0) MMA(0, 2) MMA(1, 1) MMA(1, 0) MMA(0, 1) MMA(0, 0)
#undef MMA
            if (IL && t + 1 < 16) {   // one MFMA, three VALU, ... : both kinds are available to the SIMD's arbiter at any time
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = 64 * wave + 32 * m + 8 * g + 4 * kg;
                const f32x4 bv = *(const f32x4*)(bias + (size_t)(l % nlayer_w) * 256 + f0);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[m][n][4 * g + e] + bv[e];
                        o[e] = v > 0.f ? v : v * 0.01f;
                    }
                    *(f32x4*)(X + (32 * n + j) * LDF + f0) = o;
                }
            }
        __syncthreads();
    }
    if (blockIdx.x == 0)
        for (int e = tid; e < ROWS * K; e += 256) out[e] = X[(e >> 8) * LDF + (e & 255)];
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
    kf<2, false><<<1, 256>>>(dW, dB, dX, dO, 2, NLW);
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
    float ms;
#define RUN(WGS, IL, label)                                                                                                       \
    kf<WGS, IL><<<256 * WGS, 256>>>(dW, dB, dX, dO, layers, NLW);                                                                   \
    (void)hipEventRecord(e0); kf<WGS, IL><<<256 * WGS, 256>>>(dW, dB, dX, dO, layers, NLW); (void)hipEventRecord(e1);               \
    (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);                                                          \
    printf(label ": %.3f ms for %d layers -> %.1f fp32-equivalent TFLOP/s\n", ms, layers, 256.0 * WGS * layers * 2.0 * 64 * 256 * 256 / (ms * 1e-3) / 1e12);
    RUN(1, false, "one workgroup per CU, split before the MFMAs ")
    RUN(2, false, "two workgroups per CU, split before the MFMAs")
    RUN(1, true, "one workgroup per CU, split interleaved      ")
    RUN(2, true, "two workgroups per CU, split interleaved     ")
    return 0;
}
