// Weight-gradient GEMM with a DEVICE-side row count:  dW[256, C] += G[rows, 256]^T * A[rows, C],  rows = *n_rows.
//
// The reference gets these from autograd's AddmmBackward (one cuBLAS GEMM per nn.Linear of F_color / R,
// spurfies/model/pointneus_disent.py:76-107).  Here the K dimension (rows = valid pairs or points) is only known on
// the device, so a library GEMM would force a host round trip per step; this kernel reads the count itself, which
// keeps the whole optimisation step free of host synchronisation (and capturable in a hipGraph).
//
// K-huge / MN-tiny shape: every workgroup owns a slice of the rows and the FULL 256 x C output in accumulators, streams G and A
// from HBM exactly once through an LDS ring filled by LDS-DMA and leaves its partial in a slab; a second small kernel sums the
// slabs (16 slices, float atomics) into dW.  Kernels in this file, newest first:
//   wgrad_split8_kernel   C > 32: fp32-class products from 16-bit pieces on the matrix pipe — three bf16 pieces per operand (SPF_ARITH_SPLIT, <= 2 ulp per product)
//                         or two block-scaled fp16 pieces (SPF_ARITH_H2, round 6: what the Python layer passes by default)
//   wgrad_dma_kernel      fp32 MFMA, LDS-DMA staged (arith = SPF_ARITH_F32; the verification twin of the above)
//   wgrad_lds_kernel      fp32 MFMA, register-staged (C in (128, 256) that is not 256)
//   wgrad_narrow_kernel   fp32 MFMA, direct loads (C <= 32)
#include "mlp_tile.h"

namespace {
constexpr long long COLSUM_OFF = 256LL * 256 * 256;      // workspace: [256 workgroups][256 x 256] partial slabs, then [256 workgroups][256] column sums of G
using namespace spf;

// Operand maps:
//   A operand (G^T): tile m, lane (o_l = lane & 31)  <->  output row o = 64 wave + 2 o_l + m
//   B operand (A):   tile t, lane (c_i = lane & 31)  <->  column i = col_of<NT>(c_i, t)
// col_of keeps every lane's LDS read a conflict-free 16-B access: NT = 8 -> 4 c_i + t (t < 4), 128 + 4 c_i + t - 4 (t >= 4);
// NT = 4 -> 4 c_i + t; NT = 1 -> c_i.
template <int NT>
__host__ __device__ __forceinline__ int col_of(int ci, int t) {
    return NT == 8 ? (t < 4 ? 4 * ci + t : 128 + 4 * ci + (t - 4)) : (NT == 4 ? 4 * ci + t : ci);
}

// NT in {4, 8}: the four waves of a workgroup need the SAME rows of A and different 64-column slices of G, so the rows are
// staged once per workgroup through LDS (16 rows per stage, double buffered, one barrier per 128 MFMAs per wave) instead of
// being pulled four times through the CU's vector-memory path, which is what bounded the direct-load form (~10 B/clk/CU).
template <int NT>
__global__ void __launch_bounds__(256, 1)
wgrad_lds_kernel(const float* __restrict__ G, const float* __restrict__ A, int lda, int C, const int32_t* __restrict__ n_rows_dev,
                 int max_rows, float* __restrict__ slab) {
    constexpr int U = 8, ROWS = 2 * U, CA = 32 * NT;            // rows per stage, staged A width
    constexpr int G4 = ROWS * 64 / 256, A4 = ROWS * (CA / 4) / 256;   // float4 per thread per stage
    __shared__ __attribute__((aligned(16))) float sG[2][ROWS][256];
    __shared__ __attribute__((aligned(16))) float sA[2][ROWS][CA];
    const int tid = threadIdx.x, lane = tid & 63, ci = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk += chunk & 1;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, n);
    if (r0 >= r1) return;
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    f32x4 rg[G4], ra[A4];
    auto gload = [&](int base) {
#pragma unroll
        for (int v = 0; v < G4; ++v) {
            const int e = tid + 256 * v, row = base + e / 64, c4 = e % 64;
            rg[v] = row < r1 ? *reinterpret_cast<const f32x4*>(G + (size_t)row * 256 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int v = 0; v < A4; ++v) {
            const int e = tid + 256 * v, row = base + e / (CA / 4), c4 = e % (CA / 4);
            ra[v] = (row < r1 && 4 * c4 < C) ? *reinterpret_cast<const f32x4*>(A + (size_t)row * lda + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int v = 0; v < G4; ++v) {
            const int e = tid + 256 * v;
            *reinterpret_cast<f32x4*>(&sG[buf][e / 64][4 * (e % 64)]) = rg[v];
        }
#pragma unroll
        for (int v = 0; v < A4; ++v) {
            const int e = tid + 256 * v;
            *reinterpret_cast<f32x4*>(&sA[buf][e / (CA / 4)][4 * (e % (CA / 4))]) = ra[v];
        }
    };
    gload(r0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int base = r0; base < r1; base += ROWS) {
        const bool more = base + ROWS < r1;
        if (more) gload(base + ROWS);               // next stage in flight while this one is consumed
#pragma unroll 2                                   // a fully unrolled stage needs > 512 registers at NT = 8 (spills)
        for (int u = 0; u < U; ++u) {
            const float2 a = *reinterpret_cast<const float2*>(&sG[buf][2 * u + h][64 * wave + 2 * ci]);
            float b[NT];
#pragma unroll
            for (int v = 0; v < NT / 4; ++v) {
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(&sA[buf][2 * u + h][128 * v + 4 * ci]);
                b[4 * v] = x4[0]; b[4 * v + 1] = x4[1]; b[4 * v + 2] = x4[2]; b[4 * v + 3] = x4[3];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t], acc[1][t], 0, 0, 0);
            }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = slab + ((size_t)blockIdx.x * 4 + wave) * (2 * NT * 16 * 64) + lane;   // slab[block][wave][m][t][reg][lane]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((m * NT + t) * 16 + r) * 64] = acc[m][t][r];
}

// C = 256 with LDS-DMA staging: every stage (16 rows of G and of A, 32 KB) is written into LDS by `global_load_lds_dwordx4`
// (one wave instruction = one 1 KB row, no staging registers), three buffers deep, so that one stage is always in flight
// ACROSS the barrier: counted `s_waitcnt vmcnt(8)` + raw `s_barrier`, never `__syncthreads()` (which would drain the DMA).
// With register staging the next stage's HBM latency (~2 us under load) was exposed behind a 1.7 us compute phase.
// The DMA is issued from inline asm on purpose: for the builtin, hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
// first LDS read that follows ANY pending LDS-DMA, which drains the stage that is meant to stay in flight.  M0 carries the
// wave-uniform LDS destination; lane l's 16 bytes land at dst + 16 l.
typedef __attribute__((address_space(3))) float* lptr_t;
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_row) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds_row);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// same with a wave-uniform base in SGPRs (built from kernel arguments and wave-uniform integers only, so written by the scalar
// unit: no VALU-written-SGPR hazard in front of the request) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(const float* sbase, unsigned voff, float* lds_row) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds_row);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

template <int NT>   // 8: A has 256 columns (one 1-KB row per DMA);  4: A has <= 128 columns, staged 128 wide (two rows per DMA)
__global__ void __launch_bounds__(256, NT == 8 ? 1 : 2)
wgrad_dma_kernel(const float* __restrict__ G, const float* __restrict__ A, int lda, int C, const int32_t* __restrict__ n_rows_dev,
                 int max_rows, float* __restrict__ slab) {
    constexpr int U = 8, ROWS = 2 * U, NB = 3, CA = 32 * NT;
    constexpr int NDMA = 4 + (NT == 8 ? 4 : 2);                             // DMA instructions per wave per stage
    __shared__ __attribute__((aligned(16))) float sm[NB][ROWS * (256 + CA)];  // [buffer][G 16x256 | A 16xCA]   96 / 72 KB, one object
    const int tid = threadIdx.x, lane = tid & 63, ci = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk += chunk & 1;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, n);
    if (r0 >= r1) return;
    const int nst = (r1 - r0 + ROWS - 1) / ROWS;
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    // wave w fills rows 4w..4w+3 of G and of A.  Rows past r1 re-read row r1-1 (finite data) and are cancelled on the G operand
    // below; at NT = 4 the lanes past the last real column re-read it (those output columns are dropped by the reduce kernel).
    const int c4max = (C - 1) / 4;
    auto issue = [&](int st) {
        const int buf = st % NB, base = r0 + st * ROWS;
        float* sg = &sm[buf][0];
        float* sa = sg + ROWS * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = 4 * wave + j, row = min(base + lr, r1 - 1);
            glds16(G + (size_t)row * 256 + 4 * lane, sg + lr * 256);
            if (NT == 8) glds16(A + (size_t)row * lda + 4 * lane, sa + lr * CA);
        }
        if (NT == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = 4 * wave + 2 * j + h, row = min(base + lr, r1 - 1);         // lanes 0-31: row lr, lanes 32-63: row lr + 1
                glds16(A + (size_t)row * lda + 4 * min(ci, c4max), sa + (4 * wave + 2 * j) * CA);
            }
        }
    };
    issue(0);
    if (nst > 1) issue(1);
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) {                                                      // stage st landed, st+1 may still fly
            if (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                          // ... for every wave's share of it
        if (st + 2 < nst) issue(st + 2);                                       // into the buffer read in iteration st-1
        const int buf = st % NB, left = r1 - (r0 + st * ROWS);
        const float* sg = &sm[buf][0];
        const float* sa = sg + ROWS * 256;
        // operands of step u+1 are read from LDS before the MFMAs of step u are issued (explicit register double buffer)
        float2 a[2];
        f32x4 b4[2][NT / 4];
        auto rd = [&](int u, int k) {
            a[k] = *reinterpret_cast<const float2*>(sg + (2 * u + h) * 256 + 64 * wave + 2 * ci);
            if (2 * u + h >= left) a[k] = float2{0.f, 0.f};
#pragma unroll
            for (int v = 0; v < NT / 4; ++v) b4[k][v] = *reinterpret_cast<const f32x4*>(sa + (2 * u + h) * CA + 128 * v + 4 * ci);
        };
        rd(0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = u & 1;
            if (u + 1 < U) rd(u + 1, k ^ 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].x, b4[k][t >> 2][t & 3], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].y, b4[k][t >> 2][t & 3], acc[1][t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // this stage's LDS reads retired before the next barrier
    }
    float* out = slab + ((size_t)blockIdx.x * 4 + wave) * (2 * NT * 16 * 64) + lane;   // slab[block][wave][m][t][reg][lane]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((m * NT + t) * 16 + r) * 64] = acc[m][t][r];
}

// ---- fp32 products from three bf16 pieces per operand ------------------------------------------------------------------------
// x = p1 + p2 + p3 with p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2) (both differences are exact in fp32; 3 x 8
// mantissa bits cover fp32's 24).  A product of two fp32 numbers is then sum_{i,j} a_i b_j; every a_i b_j is EXACT in the MFMA's
// fp32 accumulation (8 x 8 bits), and the three terms with i + j >= 5 are below 2^-24 of the product, so the six terms with
// i + j <= 4 reproduce the fp32 product to fp32 rounding.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 2.7x the matrix rate for a GEMM whose result differs from the fp32-MFMA one only
// by summation order.  Used where both operands stream through LDS once anyway (this weight-gradient GEMM: the splitting is
// ~6 VALU operations per staged element, hidden beside the MFMAs); the tile engine of the MLP kernels keeps fp32 MFMA.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct Split8 {
    bf16x8 p1, p2, p3;
};
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
    Split8 s;
#ifdef SPLIT_NO_SPLIT
#pragma unroll
    for (int e = 0; e < 8; ++e) { s.p1[e] = (__bf16)x[e]; s.p2[e] = s.p1[e]; s.p3[e] = s.p1[e]; }
    return s;
#endif
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a = (__bf16)x[e];                 // round to nearest even (v_cvt_pk_bf16_f32)
        const float r1 = x[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        s.p1[e] = a;
        s.p2[e] = b;
        s.p3[e] = (__bf16)r2;
    }
    return s;
}

// H2 (round 6, arith = SPF_ARITH_H2): two fp16 pieces per operand, three piece products per fp32 product (h1 h1 + h1 h2 + h2 h1; the dropped h2 h2
// is 2^-22 of the product), ALL into the one accumulator: here the second piece is the UNSCALED residual x - fp16(x) (the MLP tile engine scales it by
// 2^11 and keeps a second accumulator — mlp_tile_x3.h — for which eight waves' 128-register accumulators leave no room at two waves per SIMD).  An
// unscaled residual is a normal fp16 number — the pair then carries 22 bits of x — only for |x| >= 2^-3; below that the pair keeps an ABSOLUTE
// accuracy of 2^-25.  So the operands are block-scaled by powers of two (exact, undone once on the slab):
//   A (activations: softplus outputs, encodings, features — O(1))  x WG_H2_ASCALE = 4: |A| < 16376 (beyond: inf, loudly), 22 bits down to 0.03,
//     an absolute 7.5e-9 below;
//   G (pre-activation gradients: 1 / (3 R) of the loss times a compositing and an RBF weight per row — any magnitude)  PER WAVE (a wave owns 32
//     columns of G and their accumulators), chosen ONLINE: the first stage's largest |G| entry is put in [2^12, 2^13); when a later stage brings one
//     that would pass 2^15, the scale shrinks to put THAT one in [2^12, 2^13) and the wave's accumulators are multiplied by the (exact) ratio once.
//     An entry keeps 22 bits while it is within 2^-15 of its block's largest so far, and an absolute 2^-37 of that largest below.
// What a sum over 10^5 rows needs, not what a lone product gets from the bf16 x 3 kernel (arith = SPF_ARITH_SPLIT, which stays for operands whose
// large entries do not dominate the sum: tests/test_gpu_wgrad.py holds both to the fp32-MFMA kernel's accuracy against float64).
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
constexpr float WG_H2_ASCALE = 4.0f;
template <bool H2>
__device__ __forceinline__ Split8 split8x(const float (&x)[8], float scale) {
    if constexpr (H2) {
        Split8 s;
        f16x8w a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xs = x[e] * scale;
            const _Float16 h = (_Float16)xs;
            a[e] = h;
            b[e] = (_Float16)(xs - (float)h);
        }
        s.p1 = __builtin_bit_cast(bf16x8, a);
        s.p2 = __builtin_bit_cast(bf16x8, b);
        s.p3 = s.p1;
        return s;
    } else {
        return split8(x);
    }
}

// Stage = 16 rows (one K = 16 MFMA step) of G [16,256] and A [16,CA], written into a 4-slot fp32 ring by LDS-DMA (three stages
// requested ahead).  EIGHT waves per workgroup (two per SIMD): wave w owns output rows [32w, 32w+32) x all columns (NT x 16
// accumulator registers).  The A tile feeds every wave, so it is split into its three bf16 planes ONCE per workgroup (one
// (column, k-half) item per thread) into `planes`, laid out [piece][column][k-half] x 8 bf16 so that an MFMA B fragment is one
// 16-byte LDS read; each wave splits its own 32 columns of G in registers and issues 1/8 of the DMA requests.
// GK / AK: the operand arrives as K-MAJOR BLOCKS instead of rows: block b = rows [16 b, 16 b + 16), element (b, feature f, row r) at
// ((256 b + f) 16 + r) — what the colour kernels' transposed epilogues write straight from their accumulators (a lane owns 4
// features of one row there: 16 lanes = 64 contiguous bytes per feature).  A stage (16 rows x 256 features) is then ONE contiguous
// 16-KB run: a DMA request moves 1 KB = 16 features, its 16-byte pieces permuted so that LDS holds [row quad][feature][4 rows] and
// a thread's eight k-values are two conflict-free 16-byte reads (row-major operands need eight 4-byte reads).
// GK = 2: G as K-major TILES of 64 rows, element (row, f) at ((256 (row / 64) + f) 64 + row % 64): what a row-per-lane producer
// writes in full 128-byte lines (spf_color_backward's G3); a stage's 16 rows of a feature are then half a line (64 B) per request.
// Round 2: software pipeline with ONE barrier per stage.  The planes are double buffered, so inside one barrier period a wave
// splits stage s + 1 (ring -> planes / registers) and multiplies stage s; the fp32 ring is three stages deep (one being split, two
// in flight across the barrier): 144 KB of LDS at C = 256.  Measured on the 389 k-row colour GEMM (tools/wgrad_phases.py, cycles per
// stage of wave 0 / wave 4): DMA issue 0.3 / 0.5 k, split 1.1 k, 48 MFMAs 1.7 - 2.3 k, barrier + waits 0.5 - 2 k; 5.4 k per stage
// against 3.07 k of matrix-pipe time for the SIMD's two waves.
// SPF_WGRAD_LATE_MASK (timing builds) makes the selected waves multiply FIRST and split afterwards, so that the two waves of a SIMD
// (w, w + 4 for mask 4) are in opposite phases.  That de-phased order measured no faster (same 5.4 k): a wave's VALU stream beside
// its partner's back-to-back MFMAs runs ~2.5x slower (split 2.3 - 2.7 k instead of 1.1 k) -- the partner's next MFMA waits at the
// SIMD's vector issue for the pipe and only ~2 other vector instructions get through per 32-cycle MFMA; raising the splitting
// wave's priority (s_setprio) changes nothing, and scalar idles (s_nop) after each MFMA do free the issue port (split back to
// 1.3 k) but cost the multiplying wave more than that (48 MFMAs 2.2 - 3.8 k).  Default: every wave splits first.
#ifndef SPF_WGRAD_LATE_MASK
#define SPF_WGRAD_LATE_MASK 0
#endif
// LDS of the body below, in floats: a 3-stage fp32 ring of [G 16 x 256 | A 16 x CA] + two sets of three bf16 planes of A.  Declared by the
// KERNEL and handed in, so that one kernel can hold several instantiations (the batched launch) over one allocation.
template <int NT>
constexpr int wgrad_split8_lds_floats() { return 3 * 16 * (256 + 32 * NT) + 2 * (3 * 32 * NT * 2) * 4; }

template <int NT, int GK = 0, bool AK = false, bool H2 = false>   // 8: C = 256;  4: C <= 128 (staged 128 wide, two rows per DMA request)
__device__ __forceinline__ void wgrad_split8_body(float* sm, const float* __restrict__ G, const float* __restrict__ A, int lda, int C,
                                                  const int32_t* __restrict__ n_rows_dev, int max_rows, float* __restrict__ slab,
                                                  float* __restrict__ dbias, const int bid, const int nblk, float* __restrict__ colsum = nullptr) {
    constexpr int ROWS = 16, NB = 3, CA = 32 * NT;
    constexpr int NDMA = 2 + (NT == 8 ? 2 : 1);                           // requests per wave per stage
    constexpr int PLANE = 3 * CA * 2;                                     // bf16x8 items of one plane set: [3][CA][2]
    static_assert(wgrad_split8_lds_floats<NT>() == NB * ROWS * (256 + CA) + 2 * PLANE * 4, "LDS size");
    bf16x8* planes = reinterpret_cast<bf16x8*>(sm + NB * ROWS * (256 + CA));     // [2][3][CA][2]
    const int tid = threadIdx.x, lane = tid & 63, ci = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // 0..7
    // plane item of (column, k-half): 2 column + (k-half ^ bit 3 of the column).  A 16-byte LDS access is served in groups of 16 lanes
    // (lanes {0-3, 12-15, 20-27}, ...: all of one k-half); without the swizzle their 32-byte stride uses half of the 64 banks twice
#ifdef SPF_WGRAD_NO_SWIZZLE
    const int hs = h;
#else
    const int hs = h ^ ((ci >> 3) & 1);
#endif
    const bool late = (wave & SPF_WGRAD_LATE_MASK) != 0;                   // this wave multiplies first, its SIMD partner splits first
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    static_assert(!AK || NT == 8, "tiled A: 256 columns");
    constexpr int ALIGN = GK == 2 ? 64 : ((GK || AK) ? 16 : 2);            // blocked operands: a workgroup's rows start on a block / tile
    int chunk = (n + nblk - 1) / nblk;
    chunk = (chunk + ALIGN - 1) / ALIGN * ALIGN;
    const int r0 = bid * chunk, r1 = min(r0 + chunk, n);
    if (r0 >= r1) return;
    const int nst = (r1 - r0 + ROWS - 1) / ROWS;
#ifdef SPF_CLOCK
    T_DECL
#endif
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float gsum = 0.f;
    float gs_latest = 0.f, gs_cur = 0.f;      // H2: the scale of the newest split stage's G pieces / of the accumulators (0 = nothing but zeros yet)
#ifdef SPF_TIMING      // tools/wgrad_phases.py: cycles of waves 0 (splits first) and 4 (multiplies first) per phase of the stage loop
    unsigned long long wt[10] = {}, wl = __builtin_readcyclecounter();
#define W_MARK(i) { const unsigned long long now = __builtin_readcyclecounter(); wt[i] += now - wl; wl = now; }
#else
#define W_MARK(i)
#endif
    const int c4max = (C - 1) / 4;
    const unsigned off_row = 16u * lane, off_half = 16u * min(ci, c4max);
    const unsigned off_k = (unsigned)(((lane & 15) * 16 + 4 * (lane >> 4)) * 4);      // blocked: lane = (feature in request, row quad)
    const unsigned off_k64 = (unsigned)(((lane & 15) * 64 + 4 * (lane >> 4)) * 4);    // 64-row tiles: a feature's rows are 256 B apart
    auto issue = [&](int st) {
#ifdef SPF_WGRAD_NO_DMA       // timing build (wrong results): only the first three stages are ever requested
        if (st > 2) return;
#endif
        const int buf = st % NB, base = r0 + st * ROWS;
        float* sg = sm + buf * ROWS * (256 + CA);
        float* sa = sg + ROWS * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr = 2 * wave + j, row = min(base + lr, r1 - 1);       // wave-uniform
            // blocked: request lr = features 16 lr .. 16 lr + 15 of the 16-row block base / 16 (whole blocks exist: no clamp)
            if (GK == 2) glds16_s(G + (size_t)(base >> 6) * 16384 + lr * 1024 + (base & 63), off_k64, sg + lr * 256);
            else if (GK) glds16_s(G + (size_t)(base >> 4) * 4096 + lr * 256, off_k, sg + lr * 256);
            else glds16_s(G + (size_t)row * 256, off_row, sg + lr * 256);
            if (NT == 8) {
                if (AK) glds16_s(A + (size_t)(base >> 4) * 4096 + lr * 256, off_k, sa + lr * CA);
                else glds16_s(A + (size_t)row * lda, off_row, sa + lr * CA);
            }
        }
        if (NT == 4) {                                                     // two 512-B rows per request
            const int lr = 2 * wave, row_lo = min(base + lr, r1 - 1), row_hi = min(base + lr + 1, r1 - 1);
            const unsigned dhi = (unsigned)(row_hi - row_lo) * (unsigned)lda * 4u;
            glds16_s(A + (size_t)row_lo * lda, off_half + (h ? dhi : 0u), sa + lr * CA);
        }
    };
    // stage st out of the ring: the workgroup's shared operand into plane set st & 1 (one (column, k-half) item per thread), this
    // wave's 32 columns of G into registers
    auto split_stage = [&](int st, Split8& ga) {
        const int buf = st % NB, left = r1 - (r0 + st * ROWS);
        const float* sg = sm + buf * ROWS * (256 + CA);
        const float* sa = sg + ROWS * 256;
        bf16x8* pl = planes + (st & 1) * PLANE;
        const int col = 32 * wave + ci;
        if (col < CA) {
            float x[8];
            if (AK) {
                const float* p = sa + (col >> 4) * 256 + (2 * h) * 64 + (col & 15) * 4;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = sa[(8 * h + e) * CA + col];
            }
            W_MARK(5)
            const Split8 b = split8x<H2>(x, H2 ? WG_H2_ASCALE : 1.0f);
            W_MARK(6)
            pl[(0 * CA + col) * 2 + hs] = b.p1;
            pl[(1 * CA + col) * 2 + hs] = b.p2;
            if (!H2) pl[(2 * CA + col) * 2 + hs] = b.p3;
        }
        float x[8];
        if (GK) {
            const float* p = sg + (col >> 4) * 256 + (2 * h) * 64 + (col & 15) * 4;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = sg[(8 * h + e) * 256 + col];
        }
        W_MARK(7)
        if (left < ROWS) {                              // the workgroup's last stage only: rows past the end count as zero
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (8 * h + e < left) ? x[e] : 0.f;
        }
        if constexpr (H2) {      // this wave's scale: never above what the largest entry seen so far allows
            float m = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))), fmaxf(fmaxf(fabsf(x[4]), fabsf(x[5])), fmaxf(fabsf(x[6]), fabsf(x[7]))));
            // (one compare per stage; the wave-wide maximum only when a lane's entry would leave fp16's range — or nothing but zeros came so far)
            if (__builtin_amdgcn_ballot_w64(gs_latest == 0.f || m * gs_latest >= 32768.0f) != 0) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                if (m > 0.f && m < 3.0e38f) {
                    int ex;
                    (void)frexpf(m, &ex);                                   // m = f 2^ex, f in [0.5, 1)
                    const float want = ldexpf(1.0f, 13 - max(-100, min(ex, 100)));
                    if (gs_latest == 0.f || want < gs_latest) gs_latest = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, want)));
                }
            }
            ga = split8x<true>(x, gs_latest == 0.f ? 8192.0f : gs_latest);
        } else {
            ga = split8x<false>(x, 1.0f);
        }
        gsum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));      // column sums of G = the bias gradient
        W_MARK(8)
    };
    auto mfma_stage = [&](int st, const Split8& ga, bool ahead) {
        const bf16x8* pl = planes + (st & 1) * PLANE;
        constexpr int NQ = H2 ? 2 : 3;
        bf16x8 bq[2][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) bq[0][q] = pl[(q * CA + ci) * 2 + hs];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = t & 1;
            if (t + 1 < NT) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) bq[k ^ 1][q] = pl[(q * CA + 32 * (t + 1) + ci) * 2 + hs];
            }
            if constexpr (H2) {
                if (t == 1 && ahead) issue(st + 3);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8w g1 = __builtin_bit_cast(f16x8w, ga.p1), g2 = __builtin_bit_cast(f16x8w, ga.p2);
                const f16x8w a1 = __builtin_bit_cast(f16x8w, bq[k][0]), a2 = __builtin_bit_cast(f16x8w, bq[k][1]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g2, a1, acc[t], 0, 0, 0);      // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g1, a2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g1, a1, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            // the DMA requests of stage st + 3 ride between the MFMAs (in front of the split, where nothing covers their issue, they cost
            // 0.3 - 0.5 k cycles per stage): -1.3 % / -4 % / -2 % on the three shapes
            if (t == 1 && ahead) issue(st + 3);
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p3, bq[k][0], acc[t], 0, 0, 0);      // smallest terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p1, bq[k][2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p2, bq[k][1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p2, bq[k][0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p1, bq[k][1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga.p1, bq[k][0], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue(0);
    if (nst > 1) issue(1);
    if (nst > 2) issue(2);
    if (nst > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");                 // stage 0 landed
    else if (nst > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Split8 ga, gn;
    split_stage(0, ga);
    gs_cur = gs_latest;
    gn = ga;
    if (nst > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");                     // stage 1 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // planes of stage 0 complete, stage 1 visible
    for (int st = 0; st < nst; ++st) {
        // period st: slot st % 3 was read for the last time before the barrier above (it is refilled from inside mfma_stage); stages
        // st + 1 (landed) and st + 2 hold the others
        const bool ahead = st + 3 < nst, more = st + 1 < nst;
        W_MARK(4)
        W_MARK(0)
        if (more && !late) split_stage(st + 1, gn);
        W_MARK(1)
        mfma_stage(st, ga, ahead);
        W_MARK(2)
        if (more && late) split_stage(st + 1, gn);
        W_MARK(1)
        if (H2 && more && gs_latest != gs_cur) {          // (wave-uniform) a larger entry arrived: the accumulators follow the pieces' new scale
            if (gs_cur != 0.f) {
                const float ratio = gs_latest / gs_cur;   // powers of two: exact
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] *= ratio;
            }
            gs_cur = gs_latest;
        }
        ga = gn;
        if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");                   // stage st + 2 landed (this wave's share)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W_MARK(3)
        __builtin_amdgcn_s_barrier();
    }
#ifdef SPF_TIMING
    if (lane == 0 && (wave & 3) == 0) {
        for (int i = 0; i < 10; ++i) atomicAdd(&spf_timing_buf[16 * (wave >> 2) + i], wt[i]);
        atomicAdd(&spf_timing_buf[16 * (wave >> 2) + 10], (unsigned long long)nst);
    }
#endif
    if (dbias) {
        gsum += __shfl_xor(gsum, 32);
        if (h == 0) {
            if (colsum) colsum[(size_t)bid * 256 + 32 * wave + ci] = gsum;      // deterministic mode: summed in block order by the reduce kernel
            else atomicAdd(&dbias[32 * wave + ci], gsum);
        }
    }
    const float gs_inv = (H2 && gs_cur != 0.f) ? 1.0f / (gs_cur * WG_H2_ASCALE) : 1.0f;      // (a power of two: exact)
    float* out = slab + ((size_t)bid * 8 + wave) * (NT * 16 * 64) + lane;     // slab[block][wave 8][t][reg][lane]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64] = H2 ? acc[t][r] * gs_inv : acc[t][r];
#ifdef SPF_CLOCK
    T_FLUSH
#endif
}


template <int NT, int GK = 0, bool AK = false, bool H2 = false>
__global__ void __launch_bounds__(512, 1)
wgrad_split8_kernel(const float* __restrict__ G, const float* __restrict__ A, int lda, int C, const int32_t* __restrict__ n_rows_dev,
                    int max_rows, float* __restrict__ slab, float* __restrict__ dbias, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) float sm[wgrad_split8_lds_floats<NT>()];
    wgrad_split8_body<NT, GK, AK, H2>(sm, G, A, lda, C, n_rows_dev, max_rows, slab, dbias, (int)blockIdx.x, (int)gridDim.x, colsum);
}

// Up to three problems over the same rows in ONE launch, side by side: workgroups [first[q], first[q] + nblk[q]) of the 1-D grid belong to
// problem q (shares proportional to the problems' work), each problem with its own slab.  Three operand forms are compiled in, chosen per
// problem: row-major C = 256 (the head stage: K = valid points), G in 64-row tiles x A in 16-row blocks (the colour trunk's layers 2 and 4,
// C = 256) and G in 64-row tiles x row-major A with C <= 128 (the trunk's first layer, C = 104) — round 4: the trunk's three GEMMs were three
// launches + three reduces (each paying pipeline ramp, tail and a dispatch gap; at K = 49 k pairs, 128 rays, 174 us for 80 us of work).
// Round 5: up to SIX problems, each with its OWN row count (n_rows / max_rows): the head stage's three GEMMs (K = valid points) ride in the
// colour trunk's launch (K = pairs) — one pipeline ramp / tail / slab reduce for all the 256-wide weight gradients of a step.
constexpr int WG_MAXP = 7;
struct WgradBatch {
    const float* G[WG_MAXP];
    const float* A[WG_MAXP];
    int lda[WG_MAXP];
    float* dW[WG_MAXP];
    int ldw[WG_MAXP];
    float* dbias[WG_MAXP];
    const int32_t* n_rows[WG_MAXP];
    int max_rows[WG_MAXP];
    int C[WG_MAXP], kind[WG_MAXP], first[WG_MAXP], nblk[WG_MAXP], col_rot[WG_MAXP], col_mod[WG_MAXP];     // kind: 0 = <8, rows, rows>, 1 = <8, G64, A16>, 2 = <4, G64, rows>, 3 = <4, rows, rows>
};
// Workgroups per problem, from the ACTUAL row counts (they live on the device: the host only knows the buffers' capacities, and a share cut
// from capacities left the head stage's problems — 69 % full at 128 rays against the trunk's 60 % — as the launch's long pole: 124 us for
// 100 us of work).  Every workgroup of the GEMM launch and of the reduce launch derives the same split from the same counts: shares
// proportional to rows x width (a C <= 128 problem costs 0.62 of a C = 256 one), at least one workgroup per non-empty problem, at most one
// per WGRAD_MIN_ROWS_DEV rows; the rounding remainder goes, one workgroup at a time, to the problem with the most rows per workgroup.
constexpr int WGRAD_MIN_ROWS_DEV = 128;
__device__ __forceinline__ void wg_assign(const WgradBatch& pb, int n_problems, int B, int* first, int* nblk) {
    float w[WG_MAXP];
    int cap[WG_MAXP];
    float total = 0.f;
#pragma unroll
    for (int q = 0; q < WG_MAXP; ++q) {
        w[q] = 0.f;
        cap[q] = 0;
        nblk[q] = 0;
        if (q < n_problems) {
            const int n = pb.n_rows[q] ? min(*pb.n_rows[q], pb.max_rows[q]) : pb.max_rows[q];
            w[q] = (pb.kind[q] >= 2 ? 0.62f : 1.0f) * (float)max(n, 0);
            cap[q] = (max(n, 0) + WGRAD_MIN_ROWS_DEV - 1) / WGRAD_MIN_ROWS_DEV;
            total += w[q];
        }
    }
    int used = 0;
#pragma unroll
    for (int q = 0; q < WG_MAXP; ++q)
        if (cap[q] > 0) {
            int nb = (int)((float)B * w[q] / total);
            nb = max(1, min(nb, cap[q]));
            nblk[q] = nb;
            used += nb;
        }
    for (int it = 0; it < 2 * WG_MAXP && used < B; ++it) {      // the rounding remainder (< one workgroup per problem) — and nothing when caps bind
        int best = -1;
        float load = 0.f;
#pragma unroll
        for (int q = 0; q < WG_MAXP; ++q)
            if (nblk[q] > 0 && nblk[q] < cap[q] && w[q] / (float)nblk[q] > load) {
                load = w[q] / (float)nblk[q];
                best = q;
            }
        if (best < 0) break;
#pragma unroll
        for (int q = 0; q < WG_MAXP; ++q)
            if (q == best) ++nblk[q];
        ++used;
    }
    while (used > B) {                                          // (minimum shares of tiny problems pushed the sum over the grid: take from the largest)
        int best = 0;
#pragma unroll
        for (int q = 1; q < WG_MAXP; ++q)
            if (nblk[q] > nblk[best]) best = q;
#pragma unroll
        for (int q = 0; q < WG_MAXP; ++q)
            if (q == best) --nblk[q];
        --used;
    }
    int f = 0;
#pragma unroll
    for (int q = 0; q < WG_MAXP; ++q) {
        first[q] = f;
        f += nblk[q];
    }
}

template <bool H2>
__global__ void __launch_bounds__(512, 1)
wgrad_split8_batched_kernel(WgradBatch pb, int n_problems, float* __restrict__ slabs, size_t slab_floats, int det) {
    __shared__ __attribute__((aligned(16))) float sm[wgrad_split8_lds_floats<8>()];
    __shared__ int s_first[WG_MAXP], s_nblk[WG_MAXP];
    if (threadIdx.x == 0) wg_assign(pb, n_problems, (int)gridDim.x, s_first, s_nblk);
    __syncthreads();
    int q = -1;
#pragma unroll
    for (int t = 0; t < WG_MAXP; ++t)
        if (t < n_problems && (int)blockIdx.x >= s_first[t] && (int)blockIdx.x < s_first[t] + s_nblk[t]) q = t;
    if (q < 0) return;                                          // more workgroups than the rows need
    const int bid = (int)blockIdx.x - s_first[q], nblk = s_nblk[q];
    const int32_t* __restrict__ n_rows_dev = pb.n_rows[q];
    const int max_rows = pb.max_rows[q];
    float* slab = slabs + (size_t)q * slab_floats;
    float* colsum = det ? slab + COLSUM_OFF : nullptr;
    if (pb.kind[q] == 0) wgrad_split8_body<8, 0, false, H2>(sm, pb.G[q], pb.A[q], pb.lda[q], 256, n_rows_dev, max_rows, slab, pb.dbias[q], bid, nblk, colsum);
    else if (pb.kind[q] == 1) wgrad_split8_body<8, 2, true, H2>(sm, pb.G[q], pb.A[q], pb.lda[q], 256, n_rows_dev, max_rows, slab, pb.dbias[q], bid, nblk, colsum);
    else if (pb.kind[q] == 2) wgrad_split8_body<4, 2, false, H2>(sm, pb.G[q], pb.A[q], pb.lda[q], pb.C[q], n_rows_dev, max_rows, slab, pb.dbias[q], bid, nblk, colsum);
    else wgrad_split8_body<4, 0, false, H2>(sm, pb.G[q], pb.A[q], pb.lda[q], pb.C[q], n_rows_dev, max_rows, slab, pb.dbias[q], bid, nblk, colsum);      // row-major, C <= 128 (R.0's view-encoding columns)
}

// slab of wgrad_split8_kernel: output row o = 32 wave + C-row(reg, lane), column = 32 t + (lane & 31)
template <int NT>
__device__ __forceinline__ void wgrad_split8_reduce_body(const float* __restrict__ slab, int nblk_launched, const int32_t* __restrict__ n_rows_dev,
                                                         int max_rows, int C, float* __restrict__ dW, int ldw, int align = 2, int col_rot = 0,
                                                         int col_mod = 0, float* __restrict__ dbias_det = nullptr) {
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    if (n <= 0) return;
    int chunk = (n + nblk_launched - 1) / nblk_launched;
    chunk = (chunk + align - 1) / align * align;        // the row split of wgrad_split8_body
    const int active = (n + chunk - 1) / chunk;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (dbias_det && blockIdx.y == 0 && e < 256) {      // deterministic mode: the workgroups' column sums of G, added in block order
        const float* cs = slab + COLSUM_OFF;
        float s = 0.f;
        for (int b = 0; b < active; ++b) s += cs[(size_t)b * 256 + e];
        dbias_det[e] += s;
    }
    constexpr int PER = 8 * NT * 16 * 64;
    if (e >= PER) return;
    const int lane = e & 63, r = (e >> 6) & 15, t = (e >> 10) % NT, wave = (e >> 10) / NT;
    const int o = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int i = 32 * t + (lane & 31);
    if (i >= C) return;
    if (col_mod > 0) {                                   // product column i lands in output column (i + col_rot) mod col_mod; columns >= col_mod are padding
        if (i >= col_mod) return;
        i += col_rot;
        if (i >= col_mod) i -= col_mod;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int step = gridDim.y;
    int b = blockIdx.y;
    for (; b + 3 * step < active; b += 4 * step) {
        s0 += slab[(size_t)b * PER + e];
        s1 += slab[(size_t)(b + step) * PER + e];
        s2 += slab[(size_t)(b + 2 * step) * PER + e];
        s3 += slab[(size_t)(b + 3 * step) * PER + e];
    }
    for (; b < active; b += step) s0 += slab[(size_t)b * PER + e];
    atomicAdd(&dW[(size_t)o * ldw + i], (s0 + s1) + (s2 + s3));
}

template <int NT>
__global__ void wgrad_split8_reduce_kernel(const float* __restrict__ slab, int nblk_launched, const int32_t* __restrict__ n_rows_dev, int max_rows,
                                           int C, float* __restrict__ dW, int ldw, int align, int col_rot, int col_mod, float* __restrict__ dbias_det) {
    wgrad_split8_reduce_body<NT>(slab, nblk_launched, n_rows_dev, max_rows, C, dW, ldw, align, col_rot, col_mod, dbias_det);
}
__global__ void wgrad_split8_reduce_batched_kernel(const float* __restrict__ slabs, size_t slab_floats, WgradBatch pb, int n_problems, int n_blocks, int det) {
    __shared__ int s_first[WG_MAXP], s_nblk[WG_MAXP];
    if (threadIdx.x == 0) wg_assign(pb, n_problems, n_blocks, s_first, s_nblk);       // the GEMM launch's split, from the same counts
    __syncthreads();
    const int q = blockIdx.z;
    if (s_nblk[q] == 0) return;
    const int32_t* __restrict__ n_rows_dev = pb.n_rows[q];
    const int max_rows = pb.max_rows[q];
    const float* slab = slabs + (size_t)q * slab_floats;
    float* db = det ? pb.dbias[q] : nullptr;
    if (pb.kind[q] >= 2) wgrad_split8_reduce_body<4>(slab, s_nblk[q], n_rows_dev, max_rows, pb.C[q], pb.dW[q], pb.ldw[q], pb.kind[q] == 2 ? 64 : 2, pb.col_rot[q], pb.col_mod[q], db);
    else wgrad_split8_reduce_body<8>(slab, s_nblk[q], n_rows_dev, max_rows, 256, pb.dW[q], pb.ldw[q], pb.kind[q] == 1 ? 64 : 2, pb.col_rot[q], pb.col_mod[q], db);
}

// NT = 1 (C <= 32: the 21 view-encoding columns, a ones column for a bias gradient): direct loads, nothing to share
__global__ void __launch_bounds__(256, 1)
wgrad_narrow_kernel(const float* __restrict__ G, const float* __restrict__ A, int lda, int C, const int32_t* __restrict__ n_rows_dev,
                    int max_rows, float* __restrict__ slab) {
    const int lane = threadIdx.x & 63, ci = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    chunk += chunk & 1;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, n);
    if (r0 >= r1) return;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* gp = G + 64 * wave + 2 * ci;
    const bool bok = ci < C;
    constexpr int U = 8;
    for (int base = r0; base < r1; base += 2 * U) {
        float2 a[U];
        float b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = base + 2 * u + h;
            const bool ok = row < r1;
            const size_t rr = ok ? (size_t)row : (size_t)r0;
            a[u] = *reinterpret_cast<const float2*>(gp + rr * 256);
            if (!ok) a[u] = make_float2(0.f, 0.f);
            b[u] = (ok && bok) ? A[rr * lda + ci] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u], acc[1], 0, 0, 0);
        }
    }
    float* out = slab + ((size_t)blockIdx.x * 4 + wave) * (2 * 16 * 64) + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(m * 16 + r) * 64] = acc[m][r];
}

template <int NT>
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nblk_launched, const int32_t* __restrict__ n_rows_dev, int max_rows, int C,
                                    float* __restrict__ dW, int ldw) {
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    if (n <= 0) return;
    int chunk = (n + nblk_launched - 1) / nblk_launched;
    chunk += chunk & 1;
    const int active = (n + chunk - 1) / chunk;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // element of one slab: [wave][m][t][reg][lane]
    constexpr int PER = 4 * 2 * NT * 16 * 64;
    if (e >= PER) return;
    const int lane = e & 63, r = (e >> 6) & 15, mt = (e >> 10) % (2 * NT), wave = (e >> 10) / (2 * NT);
    const int m = mt / NT, t = mt % NT;
    const int o = 64 * wave + 2 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + m;
    const int i = col_of<NT>(lane & 31, t);
    if (i >= C) return;
    // blockIdx.y takes every gridDim.y-th slab (a serial walk over all 256 slabs is latency-bound: ~60 us whatever the width)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int step = gridDim.y;
    int b = blockIdx.y;
    for (; b + 3 * step < active; b += 4 * step) {
        s0 += slab[(size_t)b * PER + e];
        s1 += slab[(size_t)(b + step) * PER + e];
        s2 += slab[(size_t)(b + 2 * step) * PER + e];
        s3 += slab[(size_t)(b + 3 * step) * PER + e];
    }
    for (; b < active; b += step) s0 += slab[(size_t)b * PER + e];
    atomicAdd(&dW[(size_t)o * ldw + i], (s0 + s1) + (s2 + s3));
}

// dbias[c] += sum_rows G[row][c] for the kernels that do not produce it on the way (verification / narrow paths)
__global__ void colsum256_kernel(const float* __restrict__ G, const int32_t* __restrict__ n_rows_dev, int max_rows, float* __restrict__ dbias) {
    const int n = n_rows_dev ? min(*n_rows_dev, max_rows) : max_rows;
    float s = 0.f;
    for (int r = blockIdx.x; r < n; r += gridDim.x) s += G[(size_t)r * 256 + threadIdx.x];
    atomicAdd(&dbias[threadIdx.x], s);
}

}  // namespace

SPF_DEFINE_TIMING_ENTRY(spf_debug_timing_wgrad)

extern "C" {

static constexpr int RSPLIT = 16;
static constexpr int WGRAD_MIN_ROWS = 128;       // rows per workgroup below which a launch uses fewer workgroups
// slices of the slab reduce (atomically combined): ~16 slabs per thread — 16 slices for 256 slabs; with the batched launches' 60 - 99 slabs per
// problem 16 slices were 12 k workgroups of 6 additions each (18 us, mostly dispatch)
static int reduce_slices(int nblk) { const int r = nblk / 16; return r < 1 ? 1 : (r > RSPLIT ? RSPLIT : r); }
// slabs: 256 workgroups x [256 x 256] (C > 128), 512 x [256 x 128] (two workgroups per CU), 256 x [256 x 32]
int64_t spf_wgrad_workspace_floats(int32_t C) { return COLSUM_OFF + (int64_t)256 * 256; }      // slabs, then 256 workgroups x 256 column sums

int spf_wgrad(const float* G, const float* A, int32_t lda, int32_t C, const int32_t* n_rows, int32_t max_rows, float* dW, int32_t ldw,
              float* dbias, float* workspace, int32_t layout, int32_t arith, int32_t col_rot, int32_t col_mod, void* stream) {
    const bool h2 = arith == SPF_ARITH_H2;        // the 'split' family with three fp16 piece products (everything else unchanged)
    if (h2) arith = SPF_ARITH_SPLIT;
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_wgrad: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (col_mod < 0 || col_mod > C || col_rot < 0 || (col_mod > 0 && col_rot >= col_mod) || (col_mod == 0 && col_rot != 0))
        return spf::fail(SPF_EINVAL, "spf_wgrad: need 0 <= col_rot < col_mod <= C (or both 0), got col_rot=%d col_mod=%d C=%d", col_rot, col_mod, C);
    if (col_mod > 0 && !(arith == SPF_ARITH_SPLIT && C > 32)) return spf::fail(SPF_EINVAL, "spf_wgrad: the column rotation needs SPF_ARITH_SPLIT and C > 32");
    if (max_rows < 0 || C < 1 || C > 256 || lda < C || ldw < (col_mod > 0 ? col_mod : C))
        return spf::fail(SPF_EINVAL, "spf_wgrad: need 1 <= C <= 256, lda >= C, ldw >= C (col_mod with a column rotation)");
    if (max_rows == 0) return SPF_OK;
    if (!G || !A || !dW || !workspace) return spf::fail(SPF_EINVAL, "spf_wgrad: null pointer");
    const int NT = C > 128 ? 8 : (C > 32 ? 4 : 1);
    const bool g64 = layout & SPF_WGRAD_G_TILES64, gk = (layout & SPF_WGRAD_G_TILES) || g64, ak = layout & SPF_WGRAD_A_TILES;
    const bool det = layout & SPF_WGRAD_DETERMINISTIC;
    if (layout & ~(SPF_WGRAD_G_TILES | SPF_WGRAD_A_TILES | SPF_WGRAD_G_TILES64 | SPF_WGRAD_DETERMINISTIC)) return spf::fail(SPF_EINVAL, "spf_wgrad: unknown layout bits %d", layout);
    if (det && arith != SPF_ARITH_SPLIT && C > 32) return spf::fail(SPF_EINVAL, "spf_wgrad: SPF_WGRAD_DETERMINISTIC needs SPF_ARITH_SPLIT for C > 32");
    if (det && dbias && C <= 32) return spf::fail(SPF_EINVAL, "spf_wgrad: SPF_WGRAD_DETERMINISTIC with dbias needs C > 32");
    const int rsplit = det ? 1 : reduce_slices(spf::div_up(max_rows, WGRAD_MIN_ROWS) < 256 ? spf::div_up(max_rows, WGRAD_MIN_ROWS) : 256);   // det: one slice = slabs summed in block order, one plain add onto dW
    float* colsum = (det && dbias) ? workspace + COLSUM_OFF : nullptr;
    float* dbias_det = det ? dbias : nullptr;
    if ((layout & SPF_WGRAD_G_TILES) && g64) return spf::fail(SPF_EINVAL, "spf_wgrad: G is either in 16-row blocks or in 64-row tiles");
    if ((gk || ak) && (arith != SPF_ARITH_SPLIT || NT < 4 || (max_rows % (g64 ? 64 : 16)) || (ak && C != 256)))
        return spf::fail(SPF_EINVAL, "spf_wgrad: blocked operands need SPF_ARITH_SPLIT, C > 32, max_rows a multiple of 16 (64 for SPF_WGRAD_G_TILES64: whole "
                                     "blocks) and C = 256 for a blocked A");
    if (NT >= 4 && ((C % 4) || (lda % 4))) return spf::fail(SPF_EINVAL, "spf_wgrad: C and lda must be multiples of 4 for C > 32 (C=%d lda=%d)", C, lda);
    hipStream_t s = (hipStream_t)stream;
    // >= 128 rows (8 stages) per workgroup: with 512 a small-K call (the head stage at 128 rays: 6.5 k points) ran on 13 workgroups of 32
    // stages each, latency-bound at 55 us for 20 us of work; the extra slabs (256 KB each) are noise at that size
    int blocks = spf::div_up(max_rows, WGRAD_MIN_ROWS);
    const int cap = NT == 4 ? 512 : 256;   // NT = 4: half the accumulators, two workgroups per CU; else one per CU, one wave per SIMD
    if (blocks > cap) blocks = cap;
    const int per = 4 * 2 * NT * 16 * 64;
    const int align = g64 ? 64 : ((gk || ak) ? 16 : 2);
    if (arith == SPF_ARITH_SPLIT && NT == 8 && C == 256) {
        const int b8 = blocks > 256 ? 256 : blocks;      // one 8-wave workgroup per CU
        if (g64 && ak) { if (h2) wgrad_split8_kernel<8, 2, true, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 2, true, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else if (g64) { if (h2) wgrad_split8_kernel<8, 2, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 2, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else if (gk && ak) { if (h2) wgrad_split8_kernel<8, 1, true, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 1, true, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else if (gk) { if (h2) wgrad_split8_kernel<8, 1, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 1, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else if (ak) { if (h2) wgrad_split8_kernel<8, 0, true, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 0, true, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else { if (h2) wgrad_split8_kernel<8, 0, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<8, 0, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        dbias = nullptr;
        wgrad_split8_reduce_kernel<8><<<dim3(spf::div_up(per, 256), rsplit), 256, 0, s>>>(workspace, b8, n_rows, max_rows, C, dW, ldw, align, col_rot, col_mod, dbias_det);
    } else if (arith == SPF_ARITH_SPLIT && NT == 4) {
        const int b8 = blocks > 256 ? 256 : blocks;
        if (g64) { if (h2) wgrad_split8_kernel<4, 2, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<4, 2, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else if (gk) { if (h2) wgrad_split8_kernel<4, 1, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<4, 1, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        else { if (h2) wgrad_split8_kernel<4, 0, false, true><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); else wgrad_split8_kernel<4, 0, false, false><<<b8, 512, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace, dbias, colsum); }
        dbias = nullptr;
        wgrad_split8_reduce_kernel<4><<<dim3(spf::div_up(per, 256), rsplit), 256, 0, s>>>(workspace, b8, n_rows, max_rows, C, dW, ldw, align, col_rot, col_mod, dbias_det);
    } else if (NT == 8) {
        if (C == 256) wgrad_dma_kernel<8><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        else wgrad_lds_kernel<8><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<8><<<dim3(spf::div_up(per, 256), rsplit), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    } else if (NT == 4) {
        wgrad_dma_kernel<4><<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<4><<<dim3(spf::div_up(per, 256), rsplit), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    } else {
        wgrad_narrow_kernel<<<blocks, 256, 0, s>>>(G, A, lda, C, n_rows, max_rows, workspace);
        wgrad_reduce_kernel<1><<<dim3(spf::div_up(per, 256), rsplit), 256, 0, s>>>(workspace, blocks, n_rows, max_rows, C, dW, ldw);
    }
    if (dbias) colsum256_kernel<<<512, 256, 0, s>>>(G, n_rows, max_rows, dbias);
    SPF_LAUNCH_CHECK("wgrad_kernel");
    return SPF_OK;
}


int spf_wgrad_batched(const struct spf_wgrad_problem* problems, int32_t n_problems, const int32_t* n_rows, int32_t max_rows, float* workspace,
                      int32_t arith, int32_t flags, void* stream) {
    if (flags & ~SPF_WGRAD_DETERMINISTIC) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: flags may only hold SPF_WGRAD_DETERMINISTIC");
    const int det = (flags & SPF_WGRAD_DETERMINISTIC) ? 1 : 0;
    const bool h2 = arith == SPF_ARITH_H2;
    if (h2) arith = SPF_ARITH_SPLIT;
    if (arith != SPF_ARITH_SPLIT && arith != SPF_ARITH_F32) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: arith must be SPF_ARITH_SPLIT (0), SPF_ARITH_F32 (1) or SPF_ARITH_H2 (3), got %d", arith);
    if (!problems || n_problems < 1 || n_problems > WG_MAXP || max_rows < 0) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: 1 to %d problems", WG_MAXP);
    if (!workspace) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: null workspace");
    const int64_t slab_floats = spf_wgrad_workspace_floats(256);
    const int g64a16 = SPF_WGRAD_G_TILES64 | SPF_WGRAD_A_TILES;
    int kind[WG_MAXP] = {0, 0, 0, 0, 0, 0, 0};
    int rows_of[WG_MAXP];
    const int32_t* cnt_of[WG_MAXP];
    int any_rows = 0;
    for (int q = 0; q < n_problems; ++q) {
        const spf_wgrad_problem& p = problems[q];
        const int C = p.C > 0 ? p.C : 256;
        // a problem's own row count (ABI 5) or the call's
        rows_of[q] = p.max_rows > 0 ? p.max_rows : max_rows;
        cnt_of[q] = p.max_rows > 0 ? p.n_rows : n_rows;
        if (p.max_rows < 0) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: max_rows < 0", q);
        any_rows |= rows_of[q];
        if (!p.G || !p.A || !p.dW) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: null G / A / dW", q);
        // the three operand forms the side-by-side kernel holds (anything else: call spf_wgrad)
        if (C == 256 && p.layout == 0) kind[q] = 0;
        else if (C == 256 && p.layout == g64a16) kind[q] = 1;
        else if (C > 32 && C <= 128 && (C % 4) == 0 && p.layout == SPF_WGRAD_G_TILES64) kind[q] = 2;
        else if (C >= 4 && C <= 128 && (C % 4) == 0 && p.layout == 0) kind[q] = 3;
        else return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: (C, layout) = (%d, %d) is not one of (256, 0), (256, G_TILES64 | A_TILES), "
                                          "(36..128 step 4, G_TILES64), (4..128 step 4, 0)", q, C, p.layout);
        if (kind[q] != 1 && (p.lda < C || (p.lda % 4))) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: lda >= C and a multiple of 4", q);
        // (kinds 0 and 3 are row-major on both sides: any row count)
        if ((kind[q] == 1 || kind[q] == 2) && (rows_of[q] % 64)) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: tiled operands need max_rows %% 64 == 0", q);
        if (p.col_mod < 0 || p.col_mod > C || p.col_rot < 0 || (p.col_mod > 0 && p.col_rot >= p.col_mod) || (p.col_mod == 0 && p.col_rot != 0))
            return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: need 0 <= col_rot < col_mod <= C (or both 0)", q);
        if (p.ldw < (p.col_mod > 0 ? p.col_mod : C)) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: problem %d: ldw too small", q);
    }
    if (!any_rows) return SPF_OK;
    if (arith != SPF_ARITH_SPLIT || n_problems == 1) {      // fp32-MFMA verification mode / nothing to batch: the single-problem path
        for (int q = 0; q < n_problems; ++q) {
            const spf_wgrad_problem& p = problems[q];
            const int C = p.C > 0 ? p.C : 256;
            if (arith != SPF_ARITH_SPLIT && (p.layout || p.col_mod)) return spf::fail(SPF_EINVAL, "spf_wgrad_batched: tiled operands / column rotation need SPF_ARITH_SPLIT");
            if (rows_of[q] == 0) continue;
            const int rc = spf_wgrad(p.G, p.A, p.lda, C, cnt_of[q], rows_of[q], p.dW, p.ldw, p.dbias, workspace + (size_t)q * slab_floats, p.layout | flags, h2 ? SPF_ARITH_H2 : arith,
                                     p.col_rot, p.col_mod, stream);
            if (rc != SPF_OK) return rc;
        }
        return SPF_OK;
    }
    WgradBatch pb{};
    // one 8-wave workgroup per CU over ALL problems; how many each problem gets is decided ON THE DEVICE from the actual row counts
    // (wg_assign) — the host only sizes the grid: one workgroup per WGRAD_MIN_ROWS rows of capacity, at most one per CU
    long long want = 0;
    for (int q = 0; q < n_problems; ++q) {
        const int C = problems[q].C > 0 ? problems[q].C : 256;
        pb.G[q] = problems[q].G; pb.A[q] = problems[q].A; pb.lda[q] = problems[q].lda;
        pb.dW[q] = problems[q].dW; pb.ldw[q] = problems[q].ldw; pb.dbias[q] = problems[q].dbias;
        pb.n_rows[q] = cnt_of[q]; pb.max_rows[q] = rows_of[q];
        pb.C[q] = C; pb.kind[q] = kind[q]; pb.first[q] = 0; pb.nblk[q] = 0;
        pb.col_rot[q] = problems[q].col_rot; pb.col_mod[q] = problems[q].col_mod;
        want += spf::div_up(rows_of[q], WGRAD_MIN_ROWS);
    }
    const int nblocks = want < 256 ? (want < n_problems ? n_problems : (int)want) : 256;
    hipStream_t s = (hipStream_t)stream;
    const int per = 4 * 2 * 8 * 16 * 64;
    if (h2) wgrad_split8_batched_kernel<true><<<nblocks, 512, 0, s>>>(pb, n_problems, workspace, (size_t)slab_floats, det);
    else wgrad_split8_batched_kernel<false><<<nblocks, 512, 0, s>>>(pb, n_problems, workspace, (size_t)slab_floats, det);
    wgrad_split8_reduce_batched_kernel<<<dim3(spf::div_up(per, 256), det ? 1 : reduce_slices((nblocks + n_problems - 1) / n_problems * 2), n_problems), 256, 0, s>>>(
        workspace, (size_t)slab_floats, pb, n_problems, nblocks, det);
    SPF_LAUNCH_CHECK("wgrad_split8_batched_kernel");
    return SPF_OK;
}

}  // extern "C"
