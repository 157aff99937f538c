// The voxel grid as device code sees it (shared by grid.hip and the kernels that test samples against the dilated occupancy themselves:
// sampler.hip's fused training pass).
#pragma once
#include "common.h"

struct spf_grid {
    spf_grid_config cfg;
    float cell[3];
    float origin[3];
    int32_t dims[3];
    int32_t ncell;
    int32_t n_points, n_in, n_occ, max_cell;
    int32_t* cell_start;  // [ncell+1]
    float4* sorted;       // [n_in] xyz + original index (bit pattern in .w), grouped by cell
    uint32_t* dil;        // [(ncell+63)/64*2] dilated-occupancy bitmask
    int32_t* cursor;      // [ncell] scratch
    uint32_t* stats;      // [8] device scratch: min xyz, max xyz (ordered-uint), n_in, n_occ
};

namespace spf {

struct GridDev {
    float ox, oy, oz;
    float cx, cy, cz;
    int dx, dy, dz;
    const int32_t* cell_start;
    const float4* sorted;
    const uint32_t* dil;
};

inline GridDev dev_view(const spf_grid* g) {
    GridDev d;
    d.ox = g->origin[0], d.oy = g->origin[1], d.oz = g->origin[2];
    d.cx = g->cell[0], d.cy = g->cell[1], d.cz = g->cell[2];
    d.dx = g->dims[0], d.dy = g->dims[1], d.dz = g->dims[2];
    d.cell_start = g->cell_start;
    d.sorted = g->sorted;
    d.dil = g->dil;
    return d;
}

#ifdef __HIPCC__
// cell of a position; returns false when outside the grid (also for NaN)
__device__ __forceinline__ bool cell_of(const GridDev& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    float qx = (x - g.ox) / g.cx, qy = (y - g.oy) / g.cy, qz = (z - g.oz) / g.cz;
    bool in = qx >= 0.f && qx < (float)g.dx && qy >= 0.f && qy < (float)g.dy && qz >= 0.f && qz < (float)g.dz;
    cx = (int)floorf(qx);
    cy = (int)floorf(qy);
    cz = (int)floorf(qz);
    return in;
}

__device__ __forceinline__ bool dil_hit(const GridDev& g, float x, float y, float z) {
    int cx, cy, cz;
    if (!cell_of(g, x, y, z, cx, cy, cz)) return false;
    int lin = (cx * g.dy + cy) * g.dz + cz;
    return (g.dil[lin >> 5] >> (lin & 31)) & 1u;
}
#endif

#ifdef __HIPCC__
// ---- filter_points (pointneus_disent.py:207-239) of ONE slot: t = nanmean_xyz((p - o)/d), z = t at valid slots else 0,
//      delta_j = max(z_{j+1} - z_j, 0) (z_SR := 0; delta = 0 at invalid slots), x = o + z d.  Shared by render.hip's filter_points_kernel
//      and grid.hip's one-launch compaction, which can run it on the way (one launch less per optimisation step).
struct FilterArgs {
    const float* loc;         // [R*SR,3] slot positions (spf_grid_query); NULL = no filter pass
    const float* cam_loc;     // [R,3]
    const float* ray_dirs;    // [R,3]
    float* z;                 // [R*SR]
    float* deltas;            // [R*SR]
    float* x;                 // [R*SR,3]
    int SR;
};
__device__ __forceinline__ float slot_t(const float* __restrict__ loc, const float* o, const float* d) {
    float s = 0.f;
    int n = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (loc[c] - o[c]) / d[c];
        if (v == v) {  // nanmean skips NaN (0/0 when a direction component is exactly 0)
            s += v;
            ++n;
        }
    }
    return s / (float)n;  // n == 0 -> NaN, as torch.nanmean
}
__device__ __forceinline__ void filter_slot(size_t gid, const uint8_t* __restrict__ valid, const FilterArgs& f) {
    const int SR = f.SR;
    const int r = (int)(gid / SR), s = (int)(gid % SR);
    const float o[3] = {f.cam_loc[3 * r], f.cam_loc[3 * r + 1], f.cam_loc[3 * r + 2]};
    const float d[3] = {f.ray_dirs[3 * r], f.ray_dirs[3 * r + 1], f.ray_dirs[3 * r + 2]};
    const bool v = valid[gid] != 0;
    const float t = v ? slot_t(f.loc + gid * 3, o, d) : 0.f;
    float tn = 0.f;
    if (s + 1 < SR && valid[gid + 1]) tn = slot_t(f.loc + (gid + 1) * 3, o, d);
    f.z[gid] = t;
    f.deltas[gid] = v ? fmaxf(tn - t, 0.f) : 0.f;
    f.x[gid * 3] = o[0] + t * d[0];
    f.x[gid * 3 + 1] = o[1] + t * d[1];
    f.x[gid * 3 + 2] = o[2] + t * d[2];
}
#endif

// per-pair scratch the geometry kernels leave for the per-point reduction: [w, sdf_j, d sdf_j/dx (3)]
constexpr int GEO_PT_STRIDE = 5;

}  // namespace spf
