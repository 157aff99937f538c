"""ORACLE (test infrastructure, never shipped): CPU restatement of the voxel-grid kNN.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates
----------------
`torch_knnquery.VoxelGrid` as the reference calls it:
  ctor          spurfies/model/pointneus_disent.py:45-62
  set_pointset  pointneus_disent.py:252-260, 353-361, 427-435, 627-635
  query         spurfies/model/utils.py:93-95, 118-120

PARITY WITH UPSTREAM torch_knnquery IS UNPINNED: the CUDA extension is an un-vendored,
un-pinned git submodule (/root/reference/.gitmodules:1-3, empty directory), so its source
cannot be read or run.  This file freezes the build's own specification
which both this oracle and the HIP kernels implement bit-for-bit:

  cell    = float32(voxel_size * voxel_scale)                      (per axis)
  points  outside `ranges` (inclusive box) are dropped
  origin  = min(in-range points) - cell * (kernel/2)               (float32 arithmetic)
  top     = max(in-range points) + cell * (kernel/2)
  dims    = ceil((top - origin) / cell)                            (float32, then int, >= 1)
  cell(p) = floor((p - origin) / cell)                             (float32 division)
  occupied(c) = some in-range point has cell c; dilated = kernel^3 box max-pool
  a sample is a HIT iff its cell is inside the grid and dilated-occupied
  slots of a ray = its first SR hits in ray order
  neighbours of a slot = the k in-range points with smallest (dist2, index) among
      dist2 <= radius2, dist2 = (dx*dx + dy*dy) + dz*dz in float32 (no FMA),
      radius = float32(radius_limit_scale * max(voxel_size_x, voxel_size_y)),
      radius2 = radius*radius in float32;  listed in that order, -1 padded
  a ray is valid iff one of its slots has >= 1 neighbour

Everything downstream of the op (spurfies/model/utils.py:96-113) only depends on this
neighbour *set* up to floating-point summation order.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is only an accelerator for candidate generation; results do not depend on it
    from scipy.spatial import cKDTree
except Exception:  # pragma: no cover
    cKDTree = None

F32 = np.float32


class VoxelGridOracle:
    def __init__(self, voxel_size, voxel_scale, kernel_size, max_points_per_voxel, max_occ_voxels_per_example, ranges, compat=()):
        """compat: 'truncate' / 'layered' — deterministic restatements of what upstream is BELIEVED to do (SURVEY.md Appendix B;
        unverifiable, the source is absent): a cell keeps its max_points_per_voxel lowest-index points and the grid the
        max_occ_voxels cells with the lowest first index; the search stops at the sample's own cell when that already holds k points
        within the radius."""
        self.compat = tuple(compat)
        self.voxel_size = tuple(float(v) for v in voxel_size)
        self.voxel_scale = tuple(int(v) for v in voxel_scale)
        self.kernel_size = tuple(int(v) for v in kernel_size)
        # accepted for signature parity; the frozen spec keeps every point and every cell
        self.max_points_per_voxel = int(max_points_per_voxel)
        self.max_occ_voxels = int(max_occ_voxels_per_example)
        self.ranges = np.asarray(ranges, np.float64).astype(F32)
        self.cell = np.asarray([self.voxel_size[a] * self.voxel_scale[a] for a in range(3)], np.float64).astype(F32)
        self.pts = None

    # ------------------------------------------------------------------ build
    def set_pointset(self, points, actual_num_points=None):
        """points: [N,3] or [1,N,3] float32 array-like."""
        pts = np.asarray(points, F32).reshape(-1, 3)
        if actual_num_points is not None:
            pts = pts[: int(np.asarray(actual_num_points).reshape(-1)[0])]
        self.pts = np.ascontiguousarray(pts)
        lo, hi = self.ranges[:3], self.ranges[3:]
        self.in_range = np.all((pts >= lo) & (pts <= hi), axis=1)
        self.idx_in = np.nonzero(self.in_range)[0].astype(np.int64)
        pin = pts[self.in_range]
        self._grid_geometry(pin)
        if "truncate" in self.compat and len(pin):
            self._truncate()
            pin = pts[self.in_range]
        if len(pin) == 0:
            self.dims = np.zeros(3, np.int64)
            self.origin = np.zeros(3, F32)
            self.occ = np.zeros((0, 0, 0), bool)
            self.dil = self.occ
            return
        c = self.cell_of(pin)
        c = np.minimum(c, self.dims - 1)  # cannot trigger (padding), kept for safety
        occ = np.zeros(tuple(self.dims), bool)
        occ[c[:, 0], c[:, 1], c[:, 2]] = True
        self.occ = occ
        self.dil = self._dilate(occ)
        self.tree = cKDTree(pin.astype(np.float64)) if cKDTree is not None else None

    def _grid_geometry(self, pin):
        """origin / dims from ALL in-range points (truncation drops points afterwards, the grid stays)."""
        if len(pin) == 0:
            return
        half = np.asarray([k / 2.0 for k in self.kernel_size], F32)
        pad = (self.cell * half).astype(F32)
        self.origin = (pin.min(0) - pad).astype(F32)
        top = (pin.max(0) + pad).astype(F32)
        self.dims = np.maximum(np.ceil(((top - self.origin).astype(F32) / self.cell).astype(F32)).astype(np.int64), 1)

    def _truncate(self):
        ids = self.idx_in
        c = np.minimum(self.cell_of(self.pts[ids]), self.dims - 1)
        lin = (c[:, 0] * self.dims[1] + c[:, 1]) * self.dims[2] + c[:, 2]
        order = np.lexsort((ids, lin))                       # by cell, then by point index
        lin_s, ids_s = lin[order], ids[order]
        first = np.r_[True, lin_s[1:] != lin_s[:-1]]
        start = np.maximum.accumulate(np.where(first, np.arange(len(lin_s)), 0))
        keep = (np.arange(len(lin_s)) - start) < (self.max_points_per_voxel if self.max_points_per_voxel > 0 else 1 << 60)
        if self.max_occ_voxels > 0 and first.sum() > self.max_occ_voxels:
            cell_min = ids_s[first]                          # lowest index of every occupied cell
            thr = np.sort(cell_min)[self.max_occ_voxels - 1]
            keep &= (ids_s[start] <= thr)
        kept = np.zeros(len(self.pts), bool)
        kept[ids_s[keep]] = True
        self.in_range = self.in_range & kept
        self.idx_in = np.nonzero(self.in_range)[0].astype(np.int64)

    def cell_of(self, x):
        x = np.asarray(x, F32)
        return np.floor(((x - self.origin).astype(F32) / self.cell).astype(F32)).astype(np.int64)

    def _dilate(self, occ):
        out = np.zeros_like(occ)
        hk = [k // 2 for k in self.kernel_size]
        D = occ.shape
        for dx in range(-hk[0], hk[0] + 1):
            for dy in range(-hk[1], hk[1] + 1):
                for dz in range(-hk[2], hk[2] + 1):
                    sx0, sx1 = max(0, -dx), min(D[0], D[0] - dx)
                    sy0, sy1 = max(0, -dy), min(D[1], D[1] - dy)
                    sz0, sz1 = max(0, -dz), min(D[2], D[2] - dz)
                    if sx1 <= sx0 or sy1 <= sy0 or sz1 <= sz0:
                        continue
                    out[sx0 + dx : sx1 + dx, sy0 + dy : sy1 + dy, sz0 + dz : sz1 + dz] |= occ[sx0:sx1, sy0:sy1, sz0:sz1]
        return out

    # ------------------------------------------------------------------ query
    def hit_mask(self, raypos):
        """raypos [R,D,3] -> bool [R,D]: sample lies in a dilated-occupied cell of the grid."""
        x = np.asarray(raypos, F32)
        if self.dims.prod() == 0:
            return np.zeros(x.shape[:-1], bool)
        with np.errstate(invalid="ignore", over="ignore"):
            q = ((x - self.origin).astype(F32) / self.cell).astype(F32)
            fin = np.isfinite(q).all(-1)
            c = np.floor(np.where(np.isfinite(q), q, -1.0)).astype(np.int64)
        inb = fin & np.all((c >= 0) & (c < self.dims), axis=-1)
        cc = np.clip(c, 0, self.dims - 1)
        return inb & self.dil[cc[..., 0], cc[..., 1], cc[..., 2]]

    def radius(self, radius_limit_scale):
        return F32(float(radius_limit_scale) * max(self.voxel_size[0], self.voxel_size[1]))

    def knn(self, x, k, radius):
        """Exact radius-limited kNN of positions x [M,3] among in-range points.
        Returns int32 [M,k], -1 padded, sorted by (dist2, index)."""
        x = np.asarray(x, F32).reshape(-1, 3)
        out = np.full((len(x), k), -1, np.int32)
        if len(x) == 0 or len(self.idx_in) == 0:
            return out
        rad = F32(radius)
        rad2 = F32(rad * rad)
        pin = self.pts[self.idx_in]
        if self.tree is not None:
            cands = self.tree.query_ball_point(x.astype(np.float64), float(rad) * 1.001 + 1e-6)
        else:
            cands = [np.arange(len(pin))] * len(x)
        for m in range(len(x)):
            ci = np.asarray(cands[m], np.int64)
            if len(ci) == 0:
                continue
            d = (x[m][None, :] - pin[ci]).astype(F32)
            d2 = ((d[:, 0] * d[:, 0]).astype(F32) + (d[:, 1] * d[:, 1]).astype(F32)).astype(F32)
            d2 = (d2 + (d[:, 2] * d[:, 2]).astype(F32)).astype(F32)
            keep = d2 <= rad2
            cj, ci, d2 = ci[keep], self.idx_in[ci[keep]], d2[keep]
            if "layered" in self.compat and len(ci):
                own = np.all(self.cell_of(pin[cj]) == self.cell_of(x[m]), axis=1)
                if own.sum() >= k:                           # the sample's own cell already holds k points within the radius
                    ci, d2 = ci[own], d2[own]
            order = np.lexsort((ci, d2))[:k]
            out[m, : len(order)] = ci[order]
        return out

    def query_dense(self, raypos, k, radius_limit_scale, max_shading_points_per_ray):
        """Uncompacted result: pidx int32 [R,SR,k], loc f32 [R,SR,3], slot_sample int32 [R,SR]
        (index of the sample feeding each slot, -1 if the slot is unused), ray_valid bool [R]."""
        x = np.ascontiguousarray(np.asarray(raypos, F32))
        R, D = x.shape[0], x.shape[1]
        SR = int(max_shading_points_per_ray)
        hit = self.hit_mask(x)
        order = np.cumsum(hit, axis=1) - 1
        take = hit & (order < SR)
        slot_sample = np.full((R, SR), -1, np.int32)
        rr, dd = np.nonzero(take)
        slot_sample[rr, order[rr, dd]] = dd
        loc = np.zeros((R, SR, 3), F32)
        loc[rr, order[rr, dd]] = x[rr, dd]
        pidx = np.full((R, SR, k), -1, np.int32)
        nn = self.knn(x[rr, dd], k, self.radius(radius_limit_scale))
        pidx[rr, order[rr, dd]] = nn
        ray_valid = (pidx >= 0).any(axis=(1, 2))
        return pidx, loc, slot_sample, ray_valid

    def query(self, raypos, k, radius_limit_scale, max_shading_points_per_ray):
        """The op's own return convention (B=1): (sample_pidx int32 [1,Rv,SR,k],
        sample_loc f32 [1,Rv,SR,3], ray_mask int8 [1,R])."""
        x = np.asarray(raypos, F32)
        if x.ndim == 4:
            x = x[0]
        pidx, loc, _, ray_valid = self.query_dense(x, k, radius_limit_scale, max_shading_points_per_ray)
        return pidx[ray_valid][None], loc[ray_valid][None], ray_valid.astype(np.int8)[None]


def brute_force_knn(pts, x, k, radius, ranges=None):
    """Independent O(M*N) check of `VoxelGridOracle.knn` (no tree, no cells)."""
    pts = np.asarray(pts, F32)
    x = np.asarray(x, F32).reshape(-1, 3)
    ok = np.ones(len(pts), bool)
    if ranges is not None:
        r = np.asarray(ranges, np.float64).astype(F32)
        ok = np.all((pts >= r[:3]) & (pts <= r[3:]), axis=1)
    ids = np.nonzero(ok)[0]
    rad = F32(radius)
    rad2 = F32(rad * rad)
    out = np.full((len(x), k), -1, np.int32)
    for m in range(len(x)):
        d = (x[m][None] - pts[ids]).astype(F32)
        d2 = ((d[:, 0] * d[:, 0]).astype(F32) + (d[:, 1] * d[:, 1]).astype(F32)).astype(F32)
        d2 = (d2 + (d[:, 2] * d[:, 2]).astype(F32)).astype(F32)
        keep = np.nonzero(d2 <= rad2)[0]
        order = keep[np.lexsort((ids[keep], d2[keep]))][:k]
        out[m, : len(order)] = ids[order]
    return out
