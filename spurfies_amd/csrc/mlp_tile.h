// Device helpers shared by the fused MLP kernels (geometry, colour): one tile = 64 rows, four waves,
// wave w owns output columns [64w, 64w+64) as 2x2 tiles of v_mfma_f32_32x32x2_f32; A operand in LDS
// (row stride LDA floats), B operand streamed from L2 in packed fragment order.
#pragma once
#include "common.h"

namespace spf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDA = 260;     // LDS row stride in floats (1040 B: 16-B aligned, breaks the 256-B bank period)
constexpr int T_HID = 32;    // 256/8

// Pointers into the packed weight image keep their GLOBAL address space: a generic pointer makes hipcc emit flat_load, which
// counts on both vmcnt and lgkmcnt and forces a full `s_waitcnt vmcnt(0) lgkmcnt(0)` drain in every k-step of the GEMM loop.
typedef const __attribute__((address_space(1))) float* gfp;
typedef const __attribute__((address_space(1))) f32x4* gf4p;

// Re-materialise a (wave-uniform) pointer inside the persistent tile loop: keeps hipcc from hoisting the loop-invariant
// weight / bias loads of one tile iteration out of the loop and parking them in > 100 VGPRs (spills).
__device__ __forceinline__ gfp launder(const float* p) {
    asm volatile("" : "+s"(p));
    return (gfp)p;
}

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// [64][256] tile: LDS -> HBM as 1-KiB wave stores (the accumulator layout would give 4-byte stores two rows at a time,
// and 64 separately addressed stores per lane cost registers)
template <int LD = LDA>
__device__ __forceinline__ void store_tile_256(const float* X, float* __restrict__ dst, int tid) {
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int e4 = tid + 256 * u, row = e4 >> 6, c4 = e4 & 63;
        *reinterpret_cast<f32x4*>(dst + row * 256 + 4 * c4) = *reinterpret_cast<const f32x4*>(X + row * LD + 4 * c4);
    }
}

// First B fragment of a layer (k-step 0): fetched by the PREVIOUS layer's GEMM, four k-steps before its end, so that the
// ~1 us L2 latency is not exposed at the start of every layer (the two workgroups of a CU run in phase and would both wait).
struct BFrag {
    f32x4 b0, b1;
};
__device__ __forceinline__ BFrag load_bfrag(gf4p wp, int lane) {
    BFrag f;
    f.b0 = wp[lane];
    f.b1 = wp[lane + 64];
    return f;
}

// acc[mt][nt] += X[mt*32.., :] * B  for this wave's 64 output columns.  wp: [T][2][64] float4.  LD = LDS row stride.
// `first` = this layer's k-step-0 fragment (load_bfrag(wp)); returns the k-step-0 fragment of `next_wp` (or `first`).
// HBM stores that ride inside k-step 0 of a GEMM, AFTER the weight requests of k-steps 1 and 2: vmcnt retires in issue order, so
// stores placed in front of the loop make the first weight waits sit through the stores' write acknowledgements.
//   tile: the [64][256] tile in X (this GEMM's input) is copied to HBM;  m0 / m1: two sign-bit words of the preceding epilogue.
struct Side {
    float* tile = nullptr;
    int tid = 0;
    uint32_t* m0 = nullptr;
    uint32_t* m1 = nullptr;
    uint32_t b0 = 0u, b1 = 0u;
};
__device__ __forceinline__ Side side_tile(float* tile, int tid) {
    Side s;
    s.tile = tile;
    s.tid = tid;
    return s;
}

template <int T, int LD = LDA>
__device__ __forceinline__ BFrag gemm_rows64(const float* X, gf4p wp, int lane, f32x16 (&acc)[2][2], BFrag first, gf4p next_wp,
                                             const Side& side = Side()) {
    const int i = lane & 31, h = lane >> 5;
    const float* a0p = X + i * LD + 4 * h;
    const float* a1p = a0p + 32 * LD;
    gf4p bp = wp + lane;
    // weight fragments are requested TWO k-steps ahead (a k-step is 16 MFMAs = 1024 cycles, less than a loaded L2 round trip when
    // the wave has the matrix pipe to itself), the LDS operand one k-step ahead
    f32x4 b0 = first.b0, b1 = first.b1;
    f32x4 c0 = b0, c1 = b1;
    if (T > 1) {
        c0 = bp[128];
        c1 = bp[128 + 64];
    }
    f32x4 a0 = *reinterpret_cast<const f32x4*>(a0p);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p);
    BFrag nxt = first;
    constexpr int T_PRE = T > 4 ? T - 4 : 0;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
        f32x4 d0 = c0, d1 = c1;
        if (t + 2 < T) {
            d0 = bp[(t + 2) * 128];
            d1 = bp[(t + 2) * 128 + 64];
        }
        if (t == T_PRE && next_wp) nxt = load_bfrag(next_wp, lane);
        if (t == 0) {
            if (side.tile) store_tile_256<LD>(X, side.tile, side.tid);
            if (side.m0) {
                *side.m0 = side.b0;
                *side.m1 = side.b1;
            }
        }
        f32x4 na0 = a0, na1 = a1;
        if (t + 1 < T) {
            na0 = *reinterpret_cast<const f32x4*>(a0p + 8 * (t + 1));
            na1 = *reinterpret_cast<const f32x4*>(a1p + 8 * (t + 1));
        }
        // pin the order [requests for later k-steps | this k-step's 16 MFMAs]: left alone, the scheduler sinks the requests
        // to just before their use and then has to wait for the youngest load in every k-step (vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = c0; b1 = c1;
        c0 = d0; c1 = d1;
        a0 = na0; a1 = na1;
    }
    return nxt;
}

// stand-alone form: fetches its own first fragment (latency exposed)
template <int T, int LD = LDA>
__device__ __forceinline__ void gemm_rows64(const float* X, gf4p wp, int lane, f32x16 (&acc)[2][2]) {
    gemm_rows64<T, LD>(X, wp, lane, acc, load_bfrag(wp, lane), nullptr);
}

// Workgroup barrier for kernels whose waves exchange data through LDS only: __syncthreads() also drains every outstanding global
// request of the wave (s_waitcnt vmcnt(0): weight prefetches for the next layer, tile stores, atomics) before it reaches the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
}


}  // namespace spf

// Phase timing for tools/phase_times.py (build with SPF_EXTRA_HIPCC_FLAGS=-DSPF_TIMING): thread 0 of every workgroup sums the
// shader-clock cycles it spends between consecutive marks; never compiled into the product library.
#ifdef SPF_TIMING
static __device__ unsigned long long spf_timing_buf[32];      // one per translation unit
#define SPF_DEFINE_TIMING_ENTRY(name)                                                                                       \
    extern "C" int name(unsigned long long* out32, int reset) {                                                            \
        if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(spf_timing_buf), 32 * sizeof(unsigned long long)) != hipSuccess) return -5; \
        if (reset) {                                                                                                        \
            unsigned long long z[32] = {};                                                                                  \
            if (hipMemcpyToSymbol(HIP_SYMBOL(spf_timing_buf), z, sizeof(z)) != hipSuccess) return -5;                       \
        }                                                                                                                   \
        return 0;                                                                                                           \
    }
#define T_DECL unsigned long long tacc[32] = {}; unsigned long long tlast = __builtin_readcyclecounter();
#define T_MARK(i)                                                    \
    if (tid == 0) {                                                  \
        const unsigned long long now = __builtin_readcyclecounter(); \
        tacc[i] += now - tlast;                                      \
        tlast = now;                                                 \
    }
#define T_FLUSH                                                      \
    if (tid == 0)                                                    \
        for (int i = 0; i < 32; ++i) atomicAdd(&spf_timing_buf[i], tacc[i]);
#elif defined(SPF_CLOCK)
// In-kernel clock for tools/kernel_clocks.py (-DSPF_CLOCK build, never the product library): thread 0 of every workgroup stamps the
// shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at the start and the end of the kernel;
// sum of cycles / sum of ticks x 100 MHz = the clock the chip held while the kernel ran (MI355X_MICROARCH.md, DVFS give-back (6)).
static __device__ unsigned long long spf_timing_buf[32];
#define SPF_DEFINE_TIMING_ENTRY(name)                                                                                       \
    extern "C" int name(unsigned long long* out32, int reset) {                                                            \
        if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(spf_timing_buf), 32 * sizeof(unsigned long long)) != hipSuccess) return -5; \
        if (reset) {                                                                                                        \
            unsigned long long z[32] = {};                                                                                  \
            if (hipMemcpyToSymbol(HIP_SYMBOL(spf_timing_buf), z, sizeof(z)) != hipSuccess) return -5;                       \
        }                                                                                                                   \
        return 0;                                                                                                           \
    }
#define T_DECL const unsigned long long tck0 = __builtin_amdgcn_s_memtime(), trt0 = __builtin_amdgcn_s_memrealtime();
#define T_MARK(i)
#define T_FLUSH                                                                           \
    if (threadIdx.x == 0) {                                                               \
        atomicAdd(&spf_timing_buf[0], __builtin_amdgcn_s_memtime() - tck0);               \
        atomicAdd(&spf_timing_buf[1], __builtin_amdgcn_s_memrealtime() - trt0);           \
        atomicAdd(&spf_timing_buf[2], 1ull);                                              \
    }
#else
#define T_DECL
#define T_MARK(i)
#define T_FLUSH
#define SPF_DEFINE_TIMING_ENTRY(name)
#endif
