// Voxel-grid kNN for the neural point cloud (K1-K3 of DESIGN.md).
//
// Replaces torch_knnquery.VoxelGrid (reference call sites: pointneus_disent.py:45-62, 252-260,
// 353-361, 427-435, 627-635; utils.py:93-95, 118-120).  Not a translation of the upstream CUDA
// extension (its source is absent): a static counting-sorted cell table is built ONCE per cloud
// and queries are wave-cooperative (ballot + popcount slot compaction, register top-k).
//
// Float arithmetic that decides indices (cell of a point, dist2) is written without FMA
// contraction (the library is built with -ffp-contract=off) so that it matches the frozen
// specification bit for bit.
#include <stdarg.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <new>
#include <vector>

#include "common.h"
#include "grid_dev.h"

namespace spf {

char* err_buf() {
    static thread_local char buf[512];
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace spf

namespace {

using namespace spf;

__device__ __forceinline__ uint32_t ord_enc(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord_dec(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

__global__ void bbox_kernel(const float* __restrict__ pts, int n, float lx, float ly, float lz, float hx,
                            float hy, float hz, uint32_t* stats) {
    uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        bool in = x >= lx && x <= hx && y >= ly && y <= hy && z >= lz && z <= hz;
        if (in) {
            uint32_t e[3] = {ord_enc(x), ord_enc(y), ord_enc(z)};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = min(mn[a], e[a]);
                mx[a] = max(mx[a], e[a]);
            }
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], off));
            mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], off));
        }
        cnt += __shfl_xor(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&stats[a], mn[a]);
            atomicMax(&stats[3 + a], mx[a]);
        }
        atomicAdd(&stats[6], (uint32_t)cnt);
    }
}

__device__ __forceinline__ int point_cell(const GridDev& g, float x, float y, float z) {
    int cx = (int)floorf((x - g.ox) / g.cx), cy = (int)floorf((y - g.oy) / g.cy), cz = (int)floorf((z - g.oz) / g.cz);
    cx = min(max(cx, 0), g.dx - 1);
    cy = min(max(cy, 0), g.dy - 1);
    cz = min(max(cz, 0), g.dz - 1);
    return (cx * g.dy + cy) * g.dz + cz;
}

__global__ void count_kernel(const float* __restrict__ pts, int n, GridDev g, float lx, float ly, float lz,
                             float hx, float hy, float hz, int32_t* counts /* = cell_start + 1 */, const uint8_t* __restrict__ drop) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!(x >= lx && x <= hx && y >= ly && y <= hy && z >= lz && z <= hz)) return;
    if (drop && drop[i]) return;
    atomicAdd(&counts[point_cell(g, x, y, z)], 1);
}

// in-place inclusive scan of a[0..n) by ONE block of 1024 threads (a[0] is 0 on entry for the
// cell table, which makes a = exclusive starts); also copies a[i] to copy[i] for i < n-1.
__global__ void __launch_bounds__(1024) scan_kernel(int32_t* a, int n, int32_t* copy) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        int v = i < n ? a[i] : 0;
        int s = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(s, off);
            if (lane >= off) s += t;
        }
        if (lane == 63) wsum[wid] = s;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        int carry = carry_s;
        int incl = carry + woff + s;
        if (i < n) {
            a[i] = incl;
            if (copy && i < n - 1) copy[i] = incl;
        }
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
}

__global__ void fill_kernel(const float* __restrict__ pts, int n, GridDev g, float lx, float ly, float lz,
                            float hx, float hy, float hz, int32_t* cursor, float4* sorted, const uint8_t* __restrict__ drop) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!(x >= lx && x <= hx && y >= ly && y <= hy && z >= lz && z <= hz)) return;
    if (drop && drop[i]) return;
    int pos = atomicAdd(&cursor[point_cell(g, x, y, z)], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
}

// SPF_KNN_TRUNCATE, pass 1 (one thread per cell of the full table): a cell keeps its P lowest-index points (drop[] = 1 for the
// others) and reports its lowest index (INT_MAX when empty) — the order in which upstream's one-thread-per-point kernels would
// claim cells and fill them if points arrived in index order.
__global__ void truncate_cells_kernel(GridDev g, int ncell, int P, uint8_t* __restrict__ drop, int32_t* __restrict__ cell_min) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int s = g.cell_start[c], e = g.cell_start[c + 1];
    int mn = 0x7fffffff;
    for (int j = s; j < e; ++j) {
        const int ij = __float_as_int(g.sorted[j].w);
        mn = min(mn, ij);
        if (e - s > P) {
            int rank = 0;
            for (int t = s; t < e; ++t) rank += __float_as_int(g.sorted[t].w) < ij;
            if (rank >= P) drop[ij] = 1;
        }
    }
    cell_min[c] = mn;
}

// pass 2: every point of a cell whose lowest index lies above `threshold` (the max_occ-th smallest over the occupied cells) is dropped
__global__ void drop_cells_kernel(GridDev g, int ncell, const int32_t* __restrict__ cell_min, int threshold, uint8_t* __restrict__ drop) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell || cell_min[c] <= threshold) return;
    for (int j = g.cell_start[c]; j < g.cell_start[c + 1]; ++j) drop[__float_as_int(g.sorted[j].w)] = 1;
}

// one thread per cell: dilated occupancy = any occupied cell in the kernel box; 64 cells -> one ballot
__global__ void dilate_kernel(GridDev g, int ncell, int hkx, int hky, int hkz, uint32_t* dil, uint32_t* stats) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    bool occ = false, self = false;
    if (c < ncell) {
        int cz = c % g.dz, cy = (c / g.dz) % g.dy, cx = c / (g.dz * g.dy);
        self = g.cell_start[c + 1] > g.cell_start[c];
        for (int ax = max(cx - hkx, 0); ax <= min(cx + hkx, g.dx - 1); ++ax)
            for (int ay = max(cy - hky, 0); ay <= min(cy + hky, g.dy - 1); ++ay) {
                int lo = (ax * g.dy + ay) * g.dz + max(cz - hkz, 0);
                int hi = (ax * g.dy + ay) * g.dz + min(cz + hkz, g.dz - 1);
                occ |= g.cell_start[hi + 1] > g.cell_start[lo];
            }
    }
    unsigned long long b = __ballot(occ), bs = __ballot(self);
    if ((threadIdx.x & 63) == 0 && c < ncell) {
        int w = c >> 5;  // c is a multiple of 64 here
        dil[w] = (uint32_t)b;
        dil[w + 1] = (uint32_t)(b >> 32);
        if (bs) atomicAdd(&stats[7], (uint32_t)__popcll(bs));
    }
    if (c < ncell && self) atomicMax(&stats[6], (uint32_t)(g.cell_start[c + 1] - g.cell_start[c]));      // fullest cell
}

// ---- slot assignment ------------------------------------------------------------------------
// D > 1: one wave per ray; samples are tested 64 at a time, hits get consecutive slots.
__global__ void __launch_bounds__(256) hit_slots_kernel(const float* __restrict__ raypos, int R, int D, int SR,
                                                        GridDev g, int32_t* __restrict__ slot_sample, uint8_t* __restrict__ ray_valid) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;
    if (lane == 0) ray_valid[r] = 0;          // knn_kernel raises it when one of the ray's slots finds a neighbour (round 4: no ray_valid launch)
    int count = 0;
    for (int base = 0; base < D && count < SR; base += 64) {
        int d = base + lane;
        bool hit = false;
        if (d < D) {
            const float* p = raypos + ((size_t)r * D + d) * 3;
            hit = dil_hit(g, p[0], p[1], p[2]);
        }
        unsigned long long b = __ballot(hit);
        int slot = count + __popcll(b & ((1ull << lane) - 1ull));
        if (hit && slot < SR) slot_sample[(size_t)r * SR + slot] = d;
        count += __popcll(b);
    }
    for (int s = min(count, SR) + lane; s < SR; s += 64) slot_sample[(size_t)r * SR + s] = -1;
}

// D == 1 (sampler / get_sdf_eval / pseudo / tv shape): one thread per point
__global__ void point_slots_kernel(const float* __restrict__ raypos, int R, int SR, GridDev g,
                                   int32_t* __restrict__ slot_sample) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* p = raypos + (size_t)r * 3;
    bool hit = dil_hit(g, p[0], p[1], p[2]);
    slot_sample[(size_t)r * SR] = hit ? 0 : -1;
    for (int s = 1; s < SR; ++s) slot_sample[(size_t)r * SR + s] = -1;
}

// ---- k nearest within radius ----------------------------------------------------------------
// EIGHT lanes per slot (one thread per slot is bound by the latency of ~100 dependent candidate reads: 100-170 us whether
// the launch has 10^3 or 10^5 slots).  Candidates come from the 3x3 (x,y) columns of cells around the sample, each
// column's 3 z-cells being ONE contiguous run of the sorted table (z is the fastest cell axis); lane s of the octet takes
// every 8th candidate of every run and keeps its exact top-8 by (dist2, index) — one 64-bit key, dist2 >= 0 so its float
// bits order like the value — sorted in registers; three xor-shuffle rounds of bitonic merges leave the octet's top-8 in
// every lane.  The result is the specification's top-k whatever the scan order (keys are unique).
__device__ __forceinline__ void cmpx(unsigned long long& a, unsigned long long& b) {
    const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

// LAYERED (SPF_KNN_LAYERED, what upstream is believed to do — SURVEY.md Appendix B): the sample's own cell is searched first and, if it
// already holds k points within the radius, the search stops there (closer points in the neighbouring cells are then missed).
template <bool LAYERED>
__global__ void __launch_bounds__(256) knn_kernel(const float* __restrict__ raypos, int R, int D, int SR, int k,
                                                  float rad2, GridDev g, int hkx, int hky, int hkz,
                                                  int32_t* __restrict__ slot_sample, int32_t* __restrict__ pidx,
                                                  float* __restrict__ loc, uint8_t* __restrict__ slot_valid, uint8_t* __restrict__ ray_valid1,
                                                  int inline_slots) {
    constexpr unsigned long long NONE = ~0ull;
    const size_t t8 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t gid = t8 >> 3;
    const int sub = (int)(t8 & 7);
    if (gid >= (size_t)R * SR) return;                       // whole octets leave together
    int samp;
    if (inline_slots) {      // point queries (D == 1, SR == 1): the dilated-occupancy test of point_slots_kernel, done here (one launch less per pass)
        const float* p = raypos + gid * 3;
        samp = dil_hit(g, p[0], p[1], p[2]) ? 0 : -1;
        if (sub == 0) slot_sample[gid] = samp;
    } else {
        samp = slot_sample[gid];
    }
    unsigned long long key[SPF_KMAX];
#pragma unroll
    for (int t = 0; t < SPF_KMAX; ++t) key[t] = NONE;
    float x = 0.f, y = 0.f, z = 0.f;
    if (samp >= 0) {
        const size_t r = gid / SR;
        const float* p = raypos + (r * D + samp) * 3;
        x = p[0];
        y = p[1];
        z = p[2];
        int cx, cy, cz;
        cell_of(g, x, y, z, cx, cy, cz);  // a slot's sample is always inside the grid
        // a lane's candidates of one run, four at a time: the four float4 reads are independent (issued back to back, ONE L2 round trip), then
        // inserted — a run of 3 z-cells holds ~25 candidates = ~3 per lane, so a column costs one round trip instead of three or four
        // (round 5: the kernel is a chain of dependent L2 reads, 15 - 18 us at 128 rays where the launch floor is 5)
        auto insert = [&](const float4 q) {
            const float dx = x - q.x, dy = y - q.y, dz = z - q.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            const unsigned long long kk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(q.w);
            if (d2 <= rad2 && kk < key[SPF_KMAX - 1]) {
                key[SPF_KMAX - 1] = kk;
#pragma unroll
                for (int t = SPF_KMAX - 1; t > 0; --t) cmpx(key[t - 1], key[t]);
            }
        };
        auto scan = [&](int s, int e) {
            for (int j = s + sub; j < e; j += 32) {
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = g.sorted[j + 8 * u < e ? j + 8 * u : j];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j + 8 * u < e) insert(q[u]);
            }
        };
        // merge the octet's eight sorted lists: min(mine[t], partner[7 - t]) is a bitonic sequence holding the 8 smallest of both
        auto merge = [&]() {
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                unsigned long long o[SPF_KMAX];
#pragma unroll
                for (int t = 0; t < SPF_KMAX; ++t) {
                    const unsigned lo = __shfl_xor((unsigned)key[SPF_KMAX - 1 - t], m);
                    const unsigned hi = __shfl_xor((unsigned)(key[SPF_KMAX - 1 - t] >> 32), m);
                    o[t] = ((unsigned long long)hi << 32) | lo;
                }
#pragma unroll
                for (int t = 0; t < SPF_KMAX; ++t) key[t] = key[t] < o[t] ? key[t] : o[t];
#pragma unroll
                for (int j = 4; j > 0; j >>= 1)
#pragma unroll
                    for (int t = 0; t < SPF_KMAX; ++t)
                        if ((t & j) == 0) cmpx(key[t], key[t + j]);
            }
        };
        bool done = false;
        if (LAYERED) {
            const int lin = (cx * g.dy + cy) * g.dz + cz;
            scan(g.cell_start[lin], g.cell_start[lin + 1]);
            merge();
            unsigned long long kth = key[0];
#pragma unroll
            for (int t = 1; t < SPF_KMAX; ++t) kth = (t == k - 1) ? key[t] : kth;
            done = kth != NONE;                              // the same in all eight lanes of the octet
            if (!done) {
#pragma unroll
                for (int t = 0; t < SPF_KMAX; ++t) key[t] = NONE;
            }
        }
        if (!done) {
            // the runs' bounds are requested one column ahead of their scan (two dependent reads per column otherwise)
            const int ax0 = max(cx - hkx, 0), ax1 = min(cx + hkx, g.dx - 1), ay0 = max(cy - hky, 0), ay1 = min(cy + hky, g.dy - 1);
            const int zlo = max(cz - hkz, 0), zhi = min(cz + hkz, g.dz - 1) + 1;
            int ax = ax0, ay = ay0;
            int col = (ax * g.dy + ay) * g.dz;
            int rs = g.cell_start[col + zlo], re = g.cell_start[col + zhi];
            while (true) {
                const int cs = rs, ce = re;
                int nax = ax, nay = ay + 1;
                if (nay > ay1) { nay = ay0; ++nax; }
                const bool more = nax <= ax1;
                if (more) {
                    col = (nax * g.dy + nay) * g.dz;
                    rs = g.cell_start[col + zlo];
                    re = g.cell_start[col + zhi];
                }
                scan(cs, ce);
                if (!more) break;
                ax = nax;
                ay = nay;
            }
            merge();
        }
    }
    // lane s writes neighbour s; lane 0 the slot's position and validity
    unsigned long long mine = key[0];
#pragma unroll
    for (int t = 1; t < SPF_KMAX; ++t) mine = sub == t ? key[t] : mine;
    if (sub < k) pidx[gid * k + sub] = mine == NONE ? -1 : (int32_t)(unsigned)mine;
    if (sub == 0) {
        loc[gid * 3] = x;
        loc[gid * 3 + 1] = y;
        loc[gid * 3 + 2] = z;
        slot_valid[gid] = key[0] != NONE;
        // the ray's validity = "some slot has a neighbour".  SR == 1: its only slot's.  SR > 1: hit_slots_kernel cleared it; every slot with a
        // neighbour stores the same 1 (a benign same-value race), so no pass over the slots is needed afterwards
        if (ray_valid1) {
            if (SR == 1) ray_valid1[gid] = key[0] != NONE;
            else if (key[0] != NONE) ray_valid1[gid / SR] = 1;
        }
    }
}

__global__ void ray_valid_kernel(const uint8_t* __restrict__ slot_valid, int R, int SR, uint8_t* __restrict__ ray_valid) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint8_t any = 0;
    for (int s = 0; s < SR; ++s) any |= slot_valid[(size_t)r * SR + s];
    ray_valid[r] = any;
}

// ---- compaction of valid points: ordered (ray-major) lists without a host round trip ------------------
// Two launches over chunks of 2048 slots (8 consecutive slots per thread): per-chunk counts, then every
// chunk sums the counts before it (<= a few hundred values) and writes its slice.
constexpr int CMP_PER_THREAD = 8;
constexpr int CMP_CHUNK = 256 * CMP_PER_THREAD;

__device__ __forceinline__ int block_excl_scan_256(int v, int& total, int32_t* wsum /* [4] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(s, off);
        if (lane >= off) s += t;
    }
    if (lane == 63) wsum[wid] = s;
    __syncthreads();
    int woff = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wid) woff += wsum[w];
        total += wsum[w];
    }
    return woff + s - v;
}

__global__ void __launch_bounds__(256) compact_count_kernel(const uint8_t* __restrict__ slot_valid, long long nslot,
                                                            int32_t* __restrict__ chunk_counts) {
    __shared__ int32_t wsum[4];
    const long long base = (long long)blockIdx.x * CMP_CHUNK + (long long)threadIdx.x * CMP_PER_THREAD;
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u)
        if (base + u < nslot) cnt += slot_valid[base + u] != 0;
    int total;
    block_excl_scan_256(cnt, total, wsum);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) compact_write_kernel(const uint8_t* __restrict__ slot_valid, long long nslot,
                                                            const int32_t* __restrict__ chunk_counts, int32_t* __restrict__ point_slot,
                                                            int32_t* __restrict__ slot_point, int32_t* __restrict__ n_points,
                                                            float* __restrict__ fill_sdf, float fill_value, float* __restrict__ fill_grad) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t wsum2[4];
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += chunk_counts[b];
    int chunk_base;
    block_excl_scan_256(before, chunk_base, wsum2);   // total over the block = sum of all earlier chunks
    const long long base = (long long)blockIdx.x * CMP_CHUNK + (long long)threadIdx.x * CMP_PER_THREAD;
    int cnt = 0;
    uint8_t v[CMP_PER_THREAD];
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u) {
        v[u] = (base + u < nslot) ? slot_valid[base + u] : 0;
        cnt += v[u] != 0;
    }
    int total;
    int p = chunk_base + block_excl_scan_256(cnt, total, wsum);
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u)
        if (base + u < nslot) {
            if (v[u]) {
                point_slot[p] = (int32_t)(base + u);
                slot_point[base + u] = p++;
            } else {
                slot_point[base + u] = -1;
            }
            // rows the geometry kernels never write (they only touch valid points): the reference's 1000 filler / zero gradient
            if (fill_sdf) fill_sdf[base + u] = fill_value;
            if (fill_grad) {
                fill_grad[3 * (base + u)] = 0.f;
                fill_grad[3 * (base + u) + 1] = 0.f;
                fill_grad[3 * (base + u) + 2] = 0.f;
            }
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_points = chunk_base + total;
}

// ---- pair list: rows of the MLP kernels are the VALID (point, neighbour) pairs, grouped by point ---------
// (the reference's mask_to_batch_ray_idx, utils.py:172-183).  pair_off = exclusive scan of the per-point
// neighbour counts; same two-launch chunked scan as the point compaction.
__device__ __forceinline__ int nbr_count(const int32_t* __restrict__ row, int k) {
    int c = 0;
    for (int j = 0; j < k; ++j) c += row[j] >= 0;   // -1 padding is a suffix
    return c;
}

__global__ void __launch_bounds__(256) pairs_count_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot,
                                                          const int32_t* __restrict__ n_points, int max_points, int k,
                                                          int32_t* __restrict__ chunk_counts) {
    __shared__ int32_t wsum[4];
    const int P = n_points ? min(*n_points, max_points) : max_points;
    const int base = blockIdx.x * CMP_CHUNK + threadIdx.x * CMP_PER_THREAD;
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u) {
        const int p = base + u;
        if (p < P) cnt += nbr_count(nbr + (size_t)(point_slot ? point_slot[p] : p) * k, k);
    }
    int total;
    block_excl_scan_256(cnt, total, wsum);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) pairs_write_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ point_slot,
                                                          const int32_t* __restrict__ n_points, int max_points, int k,
                                                          const int32_t* __restrict__ chunk_counts, int32_t* __restrict__ pair_off,
                                                          int32_t* __restrict__ pair_point, int32_t* __restrict__ n_pairs) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t wsum2[4];
    const int P = n_points ? min(*n_points, max_points) : max_points;
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += chunk_counts[b];
    int chunk_base;
    block_excl_scan_256(before, chunk_base, wsum2);
    const int base = blockIdx.x * CMP_CHUNK + threadIdx.x * CMP_PER_THREAD;
    int c[CMP_PER_THREAD], cnt = 0;
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u) {
        const int p = base + u;
        c[u] = p < P ? nbr_count(nbr + (size_t)(point_slot ? point_slot[p] : p) * k, k) : 0;
        cnt += c[u];
    }
    int total;
    int q = chunk_base + block_excl_scan_256(cnt, total, wsum);
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u) {
        const int p = base + u;
        if (p <= P) {
            pair_off[p] = q;              // p == P closes the list
            if (p == P) *n_pairs = q;
        }
        for (int j = 0; j < c[u]; ++j) pair_point[q + j] = p;
        q += c[u];
    }
}

// ---- spf_compact_pairs: point compaction AND pair list in two launches (spf_compact_points + spf_build_pairs take four) ----------
// Chunks of 2048 slots; pass 1 counts a chunk's valid points and their neighbours, pass 2 places them behind the chunks before it.
__global__ void __launch_bounds__(256) cp_count_kernel(const uint8_t* __restrict__ slot_valid, const int32_t* __restrict__ nbr, long long nslot,
                                                       int k, int32_t* __restrict__ chunk_counts /* [chunks][2] */) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t wsum2[4];
    const long long base = (long long)blockIdx.x * CMP_CHUNK + (long long)threadIdx.x * CMP_PER_THREAD;
    int np = 0, nq = 0;
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u)
        if (base + u < nslot && slot_valid[base + u]) {
            ++np;
            nq += nbr_count(nbr + (size_t)(base + u) * k, k);
        }
    int tp, tq;
    block_excl_scan_256(np, tp, wsum);
    block_excl_scan_256(nq, tq, wsum2);
    if (threadIdx.x == 0) {
        chunk_counts[2 * blockIdx.x] = tp;
        chunk_counts[2 * blockIdx.x + 1] = tq;
    }
}

__global__ void __launch_bounds__(256) cp_write_kernel(const uint8_t* __restrict__ slot_valid, const int32_t* __restrict__ nbr, long long nslot,
                                                       int k, const int32_t* __restrict__ chunk_counts, int32_t* __restrict__ point_slot,
                                                       int32_t* __restrict__ slot_point, int32_t* __restrict__ pair_off,
                                                       int32_t* __restrict__ pair_point, int32_t* __restrict__ counts /* [n_points, n_pairs] */,
                                                       float* __restrict__ fill_sdf, float fill_value, float* __restrict__ fill_grad,
                                                       const int32_t* __restrict__ gate) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t wsum2[4];
    int bp = 0, bq = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) {
        bp += chunk_counts[2 * b];
        bq += chunk_counts[2 * b + 1];
    }
    int base_p, base_q;
    block_excl_scan_256(bp, base_p, wsum);          // totals over the block = sums over all earlier chunks
    __syncthreads();
    block_excl_scan_256(bq, base_q, wsum2);
    __syncthreads();
    const long long base = (long long)blockIdx.x * CMP_CHUNK + (long long)threadIdx.x * CMP_PER_THREAD;
    int c[CMP_PER_THREAD], np = 0, nq = 0;
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u) {
        c[u] = -1;                                   // -1: not a valid point
        if (base + u < nslot && slot_valid[base + u]) {
            c[u] = nbr_count(nbr + (size_t)(base + u) * k, k);
            ++np;
            nq += c[u];
        }
    }
    int tp, tq;
    int p = base_p + block_excl_scan_256(np, tp, wsum);
    __syncthreads();
    int q = base_q + block_excl_scan_256(nq, tq, wsum2);
#pragma unroll
    for (int u = 0; u < CMP_PER_THREAD; ++u)
        if (base + u < nslot) {
            if (c[u] >= 0) {
                point_slot[p] = (int32_t)(base + u);
                slot_point[base + u] = p;
                pair_off[p] = q;
                for (int j = 0; j < c[u]; ++j) pair_point[q + j] = p;
                q += c[u];
                ++p;
            } else {
                slot_point[base + u] = -1;
            }
            if (fill_sdf) fill_sdf[base + u] = fill_value;
            if (fill_grad) {
                fill_grad[3 * (base + u)] = 0.f;
                fill_grad[3 * (base + u) + 1] = 0.f;
                fill_grad[3 * (base + u) + 2] = 0.f;
            }
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const bool open_ = !gate || *gate != 0;      // a closed gate reports no points / pairs: the MLP kernels behind it do nothing
        counts[0] = open_ ? base_p + tp : 0;
        counts[1] = open_ ? base_q + tq : 0;
        pair_off[base_p + tp] = base_q + tq;         // closes the list
    }
}

// ---- spf_compact_pairs in ONE launch (round 4) -----------------------------------------------------------------------------
// The two launches above exist because a chunk needs the totals of every chunk before it.  Here each chunk PUBLISHES its own totals
// and reads its predecessors' (a decoupled look-back without the chain: every block adds up all earlier aggregates itself): one 64-bit
// word per chunk, {published bit, points, pairs}, written and read with relaxed agent-scope atomics.  Value and flag travel in the same
// word, so no release / acquire fence is needed (a device-scope fence writes back and invalidates the XCD's L2 - what made "last block"
// reductions slower than a second launch).  A block only ever waits for LOWER block ids, which the dispatcher
// starts no later than itself, so the wait cannot deadlock even when the grid is not co-resident.  `sync` (chunks + 1 words) must be all
// zero on entry and is left all zero: the last block to finish its look-back clears it (word 0 counts the blocks that have).
constexpr unsigned long long CPF_PUBLISHED = 1ull << 62;
constexpr int CPF_MAX_CHUNKS = 2048;      // 2048 blocks of 256 threads are co-resident on 256 CUs; larger passes keep the two-launch form
// chunks of 512 slots (two per thread), a quarter of the two-launch form's: the pass is latency-bound (a thread's slots are read one after the
// other, then four block scans and the look-back), and at 128 rays 2048-slot chunks put the sampler pass's 16 k slots on 8 workgroups
constexpr int CPF_PER_THREAD = 2;
constexpr int CPF_CHUNK = 256 * CPF_PER_THREAD;
__global__ void __launch_bounds__(256) cp_fused_kernel(const uint8_t* __restrict__ slot_valid, const int32_t* __restrict__ nbr, long long nslot,
                                                       int k, unsigned long long* __restrict__ sync, int32_t* __restrict__ point_slot,
                                                       int32_t* __restrict__ slot_point, int32_t* __restrict__ pair_off,
                                                       int32_t* __restrict__ pair_point, int32_t* __restrict__ counts /* [n_points, n_pairs] */,
                                                       float* __restrict__ fill_sdf, float fill_value, float* __restrict__ fill_grad,
                                                       const int32_t* __restrict__ gate, FilterArgs filt) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t wsum2[4];
    __shared__ int32_t wsum3[4];
    __shared__ int32_t wsum4[4];
    __shared__ int32_t i_clear;
    if (filt.loc) {          // filter_points of this thread's slots rides along (independent of the lists; issued first so that its loads overlap the scans)
#pragma unroll
        for (int u = 0; u < CPF_PER_THREAD; ++u) {
            const long long g = (long long)blockIdx.x * CPF_CHUNK + (long long)threadIdx.x * CPF_PER_THREAD + u;
            if (g < nslot) filter_slot((size_t)g, slot_valid, filt);
        }
    }
    const long long base = (long long)blockIdx.x * CPF_CHUNK + (long long)threadIdx.x * CPF_PER_THREAD;
    int c[CPF_PER_THREAD], np = 0, nq = 0;
#pragma unroll
    for (int u = 0; u < CPF_PER_THREAD; ++u) {
        c[u] = -1;                                   // -1: not a valid point
        if (base + u < nslot && slot_valid[base + u]) {
            c[u] = nbr_count(nbr + (size_t)(base + u) * k, k);
            ++np;
            nq += c[u];
        }
    }
    int tp, tq;
    const int lp = block_excl_scan_256(np, tp, wsum);
    const int lq = block_excl_scan_256(nq, tq, wsum2);
    // The LAST block publishes nothing: nobody reads its word, so nothing would prove that its store has landed before some other block
    // (the last to finish its look-back) clears the buffer — a publish overtaken by the clear would leave a stale PUBLISHED word for the next
    // launch.  Every other block's word has been READ by block gridDim.x - 1 before that block counts itself in `sync[0]`, and the clear only
    // happens after all gridDim.x blocks have counted themselves: those stores have provably landed.
    if (threadIdx.x == 0 && blockIdx.x != gridDim.x - 1)
        __hip_atomic_store(&sync[1 + blockIdx.x], CPF_PUBLISHED | ((unsigned long long)tp << 32) | (unsigned long long)tq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    int bp = 0, bq = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) {
        unsigned long long v;
        while (!((v = __hip_atomic_load(&sync[1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & CPF_PUBLISHED)) __builtin_amdgcn_s_sleep(1);
        bp += (int)((v >> 32) & 0x3fffffffu);
        bq += (int)(v & 0xffffffffu);
    }
    int base_p, base_q;
    block_excl_scan_256(bp, base_p, wsum3);         // totals over the block = sums over all earlier chunks
    block_excl_scan_256(bq, base_q, wsum4);
    if (threadIdx.x == 0) {
        // this block has read everything it needs; the last one to say so clears the words for the next launch
        const unsigned long long done = __hip_atomic_fetch_add(&sync[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i_clear = (done == (unsigned long long)gridDim.x - 1ull) ? 1 : 0;
    }
    int p = base_p + lp, q = base_q + lq;
#pragma unroll
    for (int u = 0; u < CPF_PER_THREAD; ++u)
        if (base + u < nslot) {
            if (c[u] >= 0) {
                point_slot[p] = (int32_t)(base + u);
                slot_point[base + u] = p;
                pair_off[p] = q;
                for (int j = 0; j < c[u]; ++j) pair_point[q + j] = p;
                q += c[u];
                ++p;
            } else {
                slot_point[base + u] = -1;
            }
            if (fill_sdf) fill_sdf[base + u] = fill_value;
            if (fill_grad) {
                fill_grad[3 * (base + u)] = 0.f;
                fill_grad[3 * (base + u) + 1] = 0.f;
                fill_grad[3 * (base + u) + 2] = 0.f;
            }
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const bool open_ = !gate || *gate != 0;      // a closed gate reports no points / pairs: the MLP kernels behind it do nothing
        counts[0] = open_ ? base_p + tp : 0;
        counts[1] = open_ ? base_q + tq : 0;
        pair_off[base_p + tp] = base_q + tq;         // closes the list
    }
    __syncthreads();
    if (i_clear)
        for (int b = threadIdx.x; b <= (int)gridDim.x; b += 256) __hip_atomic_store(&sync[b], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void free_tables(spf_grid* g) {
    if (g->cell_start) (void)hipFree(g->cell_start);
    if (g->sorted) (void)hipFree(g->sorted);
    if (g->dil) (void)hipFree(g->dil);
    if (g->cursor) (void)hipFree(g->cursor);
    g->cell_start = nullptr;
    g->sorted = nullptr;
    g->dil = nullptr;
    g->cursor = nullptr;
}

// ---- mesh-extraction sweep, front end (plots.py:249-253: get_sdf_eval over a regular grid): the grid's points are GENERATED from the
//      three axes (no [M,3] array travels over PCIe or through HBM), tested against the dilated occupancy — the test every point query starts
//      with (knn_kernel's inline_slots) — and the ones that pass leave compacted (coordinates + flat index); the others get the filler now.
//      ~88 % of a mesh grid fails the test: the neighbour search, the compaction and the MLP kernels then only ever see the rest.
__global__ void __launch_bounds__(256)
sweep_hits_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs, int nx, int nz, long long first,
                  long long count, GridDev g, float* __restrict__ fill, float fill_value, float* __restrict__ pts, long long* __restrict__ idx,
                  unsigned long long* __restrict__ counter) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool hit = false;
    float x = 0.f, y = 0.f, z = 0.f;
    const long long i = first + t;
    if (t < count) {        // np.meshgrid(x, y, z) 'xy' layout, raveled: i = (iy * nx + ix) * nz + iz
        const long long row = i / nz;
        x = xs[(int)(row % nx)];
        y = ys[(int)(row / nx)];
        z = zs[(int)(i - row * nz)];
        hit = dil_hit(g, x, y, z);
        if (!hit) fill[t] = fill_value;
    }
    const unsigned long long b = __ballot(hit);
    if (!b) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)__popcll(b));
    base = __shfl(base, 0);                                   // (64-bit shuffle)
    if (hit) {
        const unsigned long long o = base + __popcll(b & ((1ull << lane) - 1ull));
        pts[3 * o] = x;
        pts[3 * o + 1] = y;
        pts[3 * o + 2] = z;
        idx[o] = i;
    }
}

}  // namespace

extern "C" {

int spf_abi_version(void) { return SPF_ABI_VERSION; }
const char* spf_last_error(void) { return spf::err_buf(); }

int spf_grid_create(const spf_grid_config* cfg, spf_grid** out) {
    if (!cfg || !out) return spf::fail(SPF_EINVAL, "spf_grid_create: null argument");
    for (int a = 0; a < 3; ++a) {
        if (!(cfg->voxel_size[a] > 0.f) || cfg->voxel_scale[a] < 1 || cfg->kernel_size[a] < 1 || (cfg->kernel_size[a] & 1) == 0)
            return spf::fail(SPF_EINVAL, "spf_grid_create: voxel_size>0, voxel_scale>=1, odd kernel_size>=1 required");
        if (!(cfg->ranges[a] < cfg->ranges[a + 3])) return spf::fail(SPF_EINVAL, "spf_grid_create: empty ranges");
    }
    spf_grid* g = new (std::nothrow) spf_grid();
    if (!g) return spf::fail(SPF_ENOMEM, "spf_grid_create: out of host memory");
    memset(g, 0, sizeof(*g));
    g->cfg = *cfg;
    for (int a = 0; a < 3; ++a) g->cell[a] = (float)((double)cfg->voxel_size[a] * (double)cfg->voxel_scale[a]);
    *out = g;
    return SPF_OK;
}

void spf_grid_destroy(spf_grid* g) {
    if (!g) return;
    free_tables(g);
    if (g->stats) (void)hipFree(g->stats);
    delete g;
}

int spf_grid_build(spf_grid* g, const float* points, int32_t n, void* stream_) {
    if (!g || (!points && n > 0) || n < 0) return spf::fail(SPF_EINVAL, "spf_grid_build: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    const float* rg = g->cfg.ranges;
    if (!g->stats) SPF_HIP_CHECK(hipMalloc(&g->stats, 8 * sizeof(uint32_t)));
    uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    SPF_HIP_CHECK(hipMemcpyAsync(g->stats, init, sizeof(init), hipMemcpyHostToDevice, stream));
    if (n > 0) {
        int blocks = std::min(spf::div_up(n, 256), 1024);
        bbox_kernel<<<blocks, 256, 0, stream>>>(points, n, rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], g->stats);
        SPF_LAUNCH_CHECK("bbox_kernel");
    }
    uint32_t host[8];
    SPF_HIP_CHECK(hipMemcpyAsync(host, g->stats, sizeof(host), hipMemcpyDeviceToHost, stream));
    SPF_HIP_CHECK(hipStreamSynchronize(stream));
    free_tables(g);
    g->n_points = n;
    g->n_in = (int32_t)host[6];
    g->n_occ = 0;
    if (g->n_in == 0) {
        g->dims[0] = g->dims[1] = g->dims[2] = 0;
        g->ncell = 0;
        g->origin[0] = g->origin[1] = g->origin[2] = 0.f;
        return SPF_OK;
    }
    long long ncell = 1;
    for (int a = 0; a < 3; ++a) {
        float half = (float)g->cfg.kernel_size[a] / 2.0f;
        float pad = g->cell[a] * half;
        float mn = ord_dec(host[a]), mx = ord_dec(host[3 + a]);
        float org = mn - pad;
        float top = mx + pad;
        float ext = top - org;
        float q = ext / g->cell[a];
        int d = (int)ceilf(q);
        g->origin[a] = org;
        g->dims[a] = d < 1 ? 1 : d;
        ncell *= g->dims[a];
    }
    if (ncell > (1ll << 30)) return spf::fail(SPF_EINVAL, "spf_grid_build: grid too large (%lld cells)", ncell);
    g->ncell = (int32_t)ncell;
    size_t dil_words = (size_t)((ncell + 63) / 64) * 2;
    SPF_HIP_CHECK(hipMalloc(&g->cell_start, (size_t)(ncell + 1) * sizeof(int32_t)));
    SPF_HIP_CHECK(hipMalloc(&g->cursor, (size_t)ncell * sizeof(int32_t)));
    SPF_HIP_CHECK(hipMalloc(&g->sorted, (size_t)g->n_in * sizeof(float4)));
    SPF_HIP_CHECK(hipMalloc(&g->dil, dil_words * sizeof(uint32_t)));
    GridDev d = dev_view(g);
    auto build_tables = [&](const uint8_t* drop) -> int {
        SPF_HIP_CHECK(hipMemsetAsync(g->cell_start, 0, (size_t)(ncell + 1) * sizeof(int32_t), stream));
        count_kernel<<<spf::div_up(n, 256), 256, 0, stream>>>(points, n, d, rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], g->cell_start + 1, drop);
        SPF_LAUNCH_CHECK("count_kernel");
        scan_kernel<<<1, 1024, 0, stream>>>(g->cell_start, (int)(ncell + 1), g->cursor);
        SPF_LAUNCH_CHECK("scan_kernel");
        fill_kernel<<<spf::div_up(n, 256), 256, 0, stream>>>(points, n, d, rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], g->cursor, g->sorted, drop);
        SPF_LAUNCH_CHECK("fill_kernel");
        return SPF_OK;
    };
    if (int rc = build_tables(nullptr)) return rc;
    if (g->cfg.compat & SPF_KNN_TRUNCATE) {
        // deterministic stand-in for upstream's per-cell / per-grid capacity limits: keep the P lowest-index points of a cell and the
        // max_occ cells with the lowest first index, then rebuild the tables without the dropped points
        const int P = g->cfg.max_points_per_voxel > 0 ? g->cfg.max_points_per_voxel : 0x7fffffff;
        uint8_t* drop = nullptr;
        int32_t* cell_min = nullptr;
        SPF_HIP_CHECK(hipMalloc(&drop, (size_t)n));
        SPF_HIP_CHECK(hipMalloc(&cell_min, (size_t)ncell * sizeof(int32_t)));
        SPF_HIP_CHECK(hipMemsetAsync(drop, 0, (size_t)n, stream));
        truncate_cells_kernel<<<spf::div_up(ncell, 256), 256, 0, stream>>>(d, (int)ncell, P, drop, cell_min);
        if (g->cfg.max_occ_voxels > 0) {
            std::vector<int32_t> mins((size_t)ncell);
            SPF_HIP_CHECK(hipMemcpyAsync(mins.data(), cell_min, (size_t)ncell * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            SPF_HIP_CHECK(hipStreamSynchronize(stream));
            std::vector<int32_t> occ;
            for (int32_t v : mins)
                if (v != 0x7fffffff) occ.push_back(v);
            if ((long long)occ.size() > (long long)g->cfg.max_occ_voxels) {
                std::nth_element(occ.begin(), occ.begin() + (g->cfg.max_occ_voxels - 1), occ.end());
                drop_cells_kernel<<<spf::div_up(ncell, 256), 256, 0, stream>>>(d, (int)ncell, cell_min, occ[g->cfg.max_occ_voxels - 1], drop);
            }
        }
        const int rc = build_tables(drop);
        SPF_HIP_CHECK(hipStreamSynchronize(stream));
        (void)hipFree(drop);
        (void)hipFree(cell_min);
        if (rc) return rc;
    }
    uint32_t zero2[2] = {0u, 0u};
    SPF_HIP_CHECK(hipMemcpyAsync(g->stats + 6, zero2, sizeof(zero2), hipMemcpyHostToDevice, stream));      // [6] fullest cell, [7] occupied cells
    dilate_kernel<<<spf::div_up(ncell, 256), 256, 0, stream>>>(d, (int)ncell, g->cfg.kernel_size[0] / 2, g->cfg.kernel_size[1] / 2,
                                                             g->cfg.kernel_size[2] / 2, g->dil, g->stats);
    SPF_LAUNCH_CHECK("dilate_kernel");
    SPF_HIP_CHECK(hipMemcpyAsync(host, g->stats, sizeof(host), hipMemcpyDeviceToHost, stream));
    SPF_HIP_CHECK(hipStreamSynchronize(stream));
    g->n_occ = (int32_t)host[7];
    g->max_cell = (int32_t)host[6];
    return SPF_OK;
}

int spf_grid_get_info(const spf_grid* g, spf_grid_info* out) {
    if (!g || !out) return spf::fail(SPF_EINVAL, "spf_grid_get_info: null argument");
    for (int a = 0; a < 3; ++a) {
        out->origin[a] = g->origin[a];
        out->cell[a] = g->cell[a];
        out->dims[a] = g->dims[a];
    }
    out->n_points = g->n_points;
    out->n_in_range = g->n_in;
    out->n_occupied = g->n_occ;
    out->max_cell_points = g->max_cell;
    return SPF_OK;
}

static int grid_query_impl(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k, float radius_limit_scale, int32_t SR, int32_t* pidx,
                           float* loc, int32_t* slot_sample, uint8_t* slot_valid, uint8_t* ray_valid, bool have_slots, void* stream_);

int spf_grid_query(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k, float radius_limit_scale,
                   int32_t SR, int32_t* pidx, float* loc, int32_t* slot_sample, uint8_t* slot_valid, uint8_t* ray_valid,
                   void* stream_) {
    return grid_query_impl(g, raypos, R, D, k, radius_limit_scale, SR, pidx, loc, slot_sample, slot_valid, ray_valid, false, stream_);
}

int spf_grid_knn(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k, float radius_limit_scale, int32_t SR,
                 const int32_t* slot_sample, int32_t* pidx, float* loc, uint8_t* slot_valid, uint8_t* ray_valid, void* stream_) {
    return grid_query_impl(g, raypos, R, D, k, radius_limit_scale, SR, pidx, loc, const_cast<int32_t*>(slot_sample), slot_valid, ray_valid, true, stream_);
}

static int grid_query_impl(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k, float radius_limit_scale, int32_t SR, int32_t* pidx,
                           float* loc, int32_t* slot_sample, uint8_t* slot_valid, uint8_t* ray_valid, bool have_slots, void* stream_) {
    if (!g) return spf::fail(SPF_EINVAL, "spf_grid_query: null grid");
    if (R < 0 || D < 1 || SR < 1 || k < 1 || k > SPF_KMAX)
        return spf::fail(SPF_EINVAL, "spf_grid_query: need R>=0, D>=1, SR>=1, 1<=k<=%d (got R=%d D=%d SR=%d k=%d)", SPF_KMAX, R, D, SR, k);
    if (R == 0) return SPF_OK;
    if (!raypos || !pidx || !loc || !slot_sample || !slot_valid || !ray_valid)
        return spf::fail(SPF_EINVAL, "spf_grid_query: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t nslot = (size_t)R * SR;
    if (g->n_in == 0) {  // built on an empty / fully out-of-range cloud: nothing is ever hit
        SPF_HIP_CHECK(hipMemsetAsync(pidx, 0xff, nslot * k * sizeof(int32_t), stream));
        SPF_HIP_CHECK(hipMemsetAsync(loc, 0, nslot * 3 * sizeof(float), stream));
        SPF_HIP_CHECK(hipMemsetAsync(slot_sample, 0xff, nslot * sizeof(int32_t), stream));
        SPF_HIP_CHECK(hipMemsetAsync(slot_valid, 0, nslot, stream));
        SPF_HIP_CHECK(hipMemsetAsync(ray_valid, 0, (size_t)R, stream));
        return SPF_OK;
    }
    float vmax = g->cfg.voxel_size[0] > g->cfg.voxel_size[1] ? g->cfg.voxel_size[0] : g->cfg.voxel_size[1];
    float rad = (float)((double)radius_limit_scale * (double)vmax);
    float rad2 = rad * rad;
    GridDev d = dev_view(g);
    const int inline_slots = (D == 1 && SR == 1 && !have_slots) ? 1 : 0;
    if (have_slots) {
        // slot_sample (and the cleared ray_valid) come from the caller: spf_sampler_train assigned the slots on its way
    } else if (inline_slots) {
        // the slot test rides inside knn_kernel
    } else if (D == 1) {
        point_slots_kernel<<<spf::div_up(R, 256), 256, 0, stream>>>(raypos, R, SR, d, slot_sample);
        SPF_LAUNCH_CHECK("point_slots_kernel");
    } else {
        hit_slots_kernel<<<spf::div_up((long long)R * 64, 256), 256, 0, stream>>>(raypos, R, D, SR, d, slot_sample, ray_valid);
        SPF_LAUNCH_CHECK("hit_slots_kernel");
    }
    if (g->cfg.compat & SPF_KNN_LAYERED)
        knn_kernel<true><<<spf::div_up((long long)nslot * 8, 256), 256, 0, stream>>>(raypos, R, D, SR, k, rad2, d, g->cfg.kernel_size[0] / 2,
                                                                                 g->cfg.kernel_size[1] / 2, g->cfg.kernel_size[2] / 2,
                                                                                 slot_sample, pidx, loc, slot_valid, ray_valid, inline_slots);
    else
        knn_kernel<false><<<spf::div_up((long long)nslot * 8, 256), 256, 0, stream>>>(raypos, R, D, SR, k, rad2, d, g->cfg.kernel_size[0] / 2,
                                                                                  g->cfg.kernel_size[1] / 2, g->cfg.kernel_size[2] / 2,
                                                                                  slot_sample, pidx, loc, slot_valid, ray_valid, inline_slots);
    SPF_LAUNCH_CHECK("knn_kernel");
    if (SR > 1 && D == 1) {      // (a shape no pass of the model uses: point queries with several slots; rays go through hit_slots + knn above)
        ray_valid_kernel<<<spf::div_up(R, 256), 256, 0, stream>>>(slot_valid, R, SR, ray_valid);
        SPF_LAUNCH_CHECK("ray_valid_kernel");
    }
    return SPF_OK;
}

int spf_compact_points(const uint8_t* slot_valid, int32_t R, int32_t SR, int32_t* point_slot, int32_t* slot_point,
                       int32_t* n_points, int32_t* scratch, float* fill_sdf, float fill_value, float* fill_grad, void* stream_) {
    if (R < 0 || SR < 1) return spf::fail(SPF_EINVAL, "spf_compact_points: bad sizes");
    if (!n_points) return spf::fail(SPF_EINVAL, "spf_compact_points: null n_points");
    hipStream_t stream = (hipStream_t)stream_;
    if (R == 0) {
        SPF_HIP_CHECK(hipMemsetAsync(n_points, 0, sizeof(int32_t), stream));
        return SPF_OK;
    }
    if (!slot_valid || !point_slot || !slot_point || !scratch) return spf::fail(SPF_EINVAL, "spf_compact_points: null buffer");
    const long long nslot = (long long)R * SR;
    const int chunks = spf::div_up(nslot, CMP_CHUNK);
    if (chunks > R + 1) return spf::fail(SPF_EINVAL, "spf_compact_points: scratch (R+1 ints) too small for %d chunks", chunks);
    compact_count_kernel<<<chunks, 256, 0, stream>>>(slot_valid, nslot, scratch);
    SPF_LAUNCH_CHECK("compact_count_kernel");
    compact_write_kernel<<<chunks, 256, 0, stream>>>(slot_valid, nslot, scratch, point_slot, slot_point, n_points, fill_sdf, fill_value, fill_grad);
    SPF_LAUNCH_CHECK("compact_write_kernel");
    return SPF_OK;
}

// load-time voxel thinning (spurfies/model/utils.py:21-27): cell = floor((p - space_min) / voxel) in float32 with IEEE division
__global__ void voxel_cells_kernel(const float* __restrict__ xyz, long long n, float mx, float my, float mz, float v, int32_t* __restrict__ cells) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cells[3 * i + 0] = (int32_t)floorf((xyz[3 * i + 0] - mx) / v);
    cells[3 * i + 1] = (int32_t)floorf((xyz[3 * i + 1] - my) / v);
    cells[3 * i + 2] = (int32_t)floorf((xyz[3 * i + 2] - mz) / v);
}

int spf_voxel_cells(const float* xyz, int64_t n, const float* space_min, float voxel, int32_t* cells, void* stream_) {
    if (n < 0 || !(voxel > 0.f) || !space_min) return spf::fail(SPF_EINVAL, "spf_voxel_cells: need n >= 0, voxel > 0, space_min");
    if (n == 0) return SPF_OK;
    if (!xyz || !cells) return spf::fail(SPF_EINVAL, "spf_voxel_cells: null buffer");
    voxel_cells_kernel<<<spf::div_up((long long)n, 256), 256, 0, (hipStream_t)stream_>>>(xyz, (long long)n, space_min[0], space_min[1], space_min[2], voxel,
                                                                                       cells);
    SPF_LAUNCH_CHECK("voxel_cells_kernel");
    return SPF_OK;
}

int spf_build_pairs(const int32_t* nbr, const int32_t* point_slot, const int32_t* n_points, int32_t max_points, int32_t k,
                    int32_t* pair_off, int32_t* pair_point, int32_t* n_pairs, int32_t* scratch, void* stream_) {
    if (max_points < 0 || k < 1 || k > SPF_KMAX) return spf::fail(SPF_EINVAL, "spf_build_pairs: bad sizes");
    if (!nbr || !pair_off || !pair_point || !n_pairs || !scratch) return spf::fail(SPF_EINVAL, "spf_build_pairs: null buffer");
    hipStream_t stream = (hipStream_t)stream_;
    const int chunks = spf::div_up((long long)max_points + 1, CMP_CHUNK);   // +1: the closing entry pair_off[P]
    pairs_count_kernel<<<chunks, 256, 0, stream>>>(nbr, point_slot, n_points, max_points, k, scratch);
    SPF_LAUNCH_CHECK("pairs_count_kernel");
    pairs_write_kernel<<<chunks, 256, 0, stream>>>(nbr, point_slot, n_points, max_points, k, scratch, pair_off, pair_point, n_pairs);
    SPF_LAUNCH_CHECK("pairs_write_kernel");
    return SPF_OK;
}

int64_t spf_compact_sync_words(int64_t n_slots) { return spf::div_up(n_slots, (int64_t)CPF_CHUNK) + 1; }

static int compact_pairs_impl(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot, int32_t* slot_point,
                              int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch, float* fill_sdf, float fill_value,
                              float* fill_grad, const int32_t* gate, uint64_t* sync, FilterArgs filt, void* stream_);

int spf_compact_pairs(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot, int32_t* slot_point,
                      int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch, float* fill_sdf, float fill_value,
                      float* fill_grad, const int32_t* gate, uint64_t* sync, void* stream_) {
    return compact_pairs_impl(slot_valid, nbr, R, SR, k, point_slot, slot_point, pair_off, pair_point, counts, scratch, fill_sdf, fill_value, fill_grad, gate, sync,
                              FilterArgs{}, stream_);
}

int spf_compact_pairs_filter(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot, int32_t* slot_point,
                             int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch, float* fill_sdf, float fill_value,
                             float* fill_grad, const int32_t* gate, uint64_t* sync, const float* loc, const float* cam_loc, const float* ray_dirs,
                             float* z, float* deltas, float* x, void* stream_) {
    if (!loc || !cam_loc || !ray_dirs || !z || !deltas || !x) return spf::fail(SPF_EINVAL, "spf_compact_pairs_filter: null filter_points buffer");
    return compact_pairs_impl(slot_valid, nbr, R, SR, k, point_slot, slot_point, pair_off, pair_point, counts, scratch, fill_sdf, fill_value, fill_grad, gate, sync,
                              FilterArgs{loc, cam_loc, ray_dirs, z, deltas, x, SR}, stream_);
}

// filter_points on its own (the two-launch form of the compaction, or passes of more than 1 M slots)
__global__ void filter_slots_kernel(const uint8_t* __restrict__ valid, long long nslot, FilterArgs filt) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < nslot) filter_slot((size_t)g, valid, filt);
}

static int compact_pairs_impl(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot, int32_t* slot_point,
                              int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch, float* fill_sdf, float fill_value,
                              float* fill_grad, const int32_t* gate, uint64_t* sync, FilterArgs filt, void* stream_) {
    if (R < 0 || SR < 1 || k < 1 || k > SPF_KMAX) return spf::fail(SPF_EINVAL, "spf_compact_pairs: bad sizes");
    if (!counts || !pair_off) return spf::fail(SPF_EINVAL, "spf_compact_pairs: null counts / pair_off");
    hipStream_t stream = (hipStream_t)stream_;
    if (R == 0) {
        SPF_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), stream));
        SPF_HIP_CHECK(hipMemsetAsync(pair_off, 0, sizeof(int32_t), stream));
        return SPF_OK;
    }
    if (!slot_valid || !nbr || !point_slot || !slot_point || !pair_point || !scratch) return spf::fail(SPF_EINVAL, "spf_compact_pairs: null buffer");
    const long long nslot = (long long)R * SR;
    const int chunks = spf::div_up(nslot, CMP_CHUNK);
    if (sync && spf::div_up(nslot, (long long)CPF_CHUNK) <= CPF_MAX_CHUNKS) {          // one launch: chunks publish their totals to each other (cp_fused_kernel)
        cp_fused_kernel<<<(int)spf::div_up(nslot, (long long)CPF_CHUNK), 256, 0, stream>>>(slot_valid, nbr, nslot, k, reinterpret_cast<unsigned long long*>(sync), point_slot, slot_point,
                                                    pair_off, pair_point, counts, fill_sdf, fill_value, fill_grad, gate, filt);
        SPF_LAUNCH_CHECK("cp_fused_kernel");
        return SPF_OK;
    }
    if (filt.loc) {
        filter_slots_kernel<<<spf::div_up(nslot, 256), 256, 0, stream>>>(slot_valid, nslot, filt);
        SPF_LAUNCH_CHECK("filter_slots_kernel");
    }
    cp_count_kernel<<<chunks, 256, 0, stream>>>(slot_valid, nbr, nslot, k, scratch);
    SPF_LAUNCH_CHECK("cp_count_kernel");
    cp_write_kernel<<<chunks, 256, 0, stream>>>(slot_valid, nbr, nslot, k, scratch, point_slot, slot_point, pair_off, pair_point, counts, fill_sdf,
                                                fill_value, fill_grad, gate);
    SPF_LAUNCH_CHECK("cp_write_kernel");
    return SPF_OK;
}

int spf_grid_sweep_hits(const spf_grid* g, const float* xs, const float* ys, const float* zs, int32_t nx, int32_t ny, int32_t nz, int64_t first,
                        int64_t count, float* fill, float fill_value, float* pts, int64_t* idx, uint64_t* counter, void* stream_) {
    if (!g || !g->cell_start) return spf::fail(SPF_EINVAL, "spf_grid_sweep_hits: grid not built");
    if (nx < 1 || ny < 1 || nz < 1 || first < 0 || count < 0 || first + count > (int64_t)nx * ny * nz)
        return spf::fail(SPF_EINVAL, "spf_grid_sweep_hits: need nx, ny, nz >= 1 and [first, first + count) inside the nx * ny * nz grid");
    if (count == 0) return SPF_OK;
    if (!xs || !ys || !zs || !fill || !pts || !idx || !counter) return spf::fail(SPF_EINVAL, "spf_grid_sweep_hits: null pointer");
    sweep_hits_kernel<<<spf::div_up(count, 256), 256, 0, (hipStream_t)stream_>>>(xs, ys, zs, nx, nz, first, count, spf::dev_view(g), fill, fill_value, pts,
                                                                                   reinterpret_cast<long long*>(idx),
                                                                                   reinterpret_cast<unsigned long long*>(counter));
    SPF_LAUNCH_CHECK("sweep_hits_kernel");
    return SPF_OK;
}

}  // extern "C"
