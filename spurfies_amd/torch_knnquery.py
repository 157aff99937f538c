"""Drop-in for the `torch_knnquery` module (the un-vendored CUDA extension the reference imports
at spurfies/model/pointneus_disent.py:9).  Same class name, constructor arguments, method names,
tensor shapes and dtypes as the reference's call sites expect:

    ctor          spurfies/model/pointneus_disent.py:45-62
    set_pointset  spurfies/model/pointneus_disent.py:252-260 (and :353, :427, :522, :627)
    query         spurfies/model/utils.py:93-95, 118-120

All compute runs in libspurfies_hip.so (spf_grid_*); this file only owns tensors.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, _prof


KNN_TRUNCATE, KNN_LAYERED = 1, 2       # include/spurfies_hip.h: SPF_KNN_*


class VoxelGrid:
    def __init__(self, voxel_size, voxel_scale, kernel_size, max_points_per_voxel, max_occ_voxels_per_example, ranges, compat=()):
        """The reference's six positional arguments (pointneus_disent.py:45-62).  `compat`: names of the upstream-compatibility
        switches to apply, 'truncate' (deterministic max_points_per_voxel / max_occ_voxels limits) and / or 'layered' (own-cell early
        exit) — what the absent CUDA source is believed to do; default: the frozen exact specification."""
        self.voxel_size = tuple(float(v) for v in voxel_size)
        self.voxel_scale = tuple(int(v) for v in voxel_scale)
        self.kernel_size = tuple(int(v) for v in kernel_size)
        self.ranges = tuple(float(v) for v in ranges)
        cfg = _lib.GridConfig()
        cfg.voxel_size[:] = self.voxel_size
        cfg.voxel_scale[:] = self.voxel_scale
        cfg.kernel_size[:] = self.kernel_size
        cfg.max_points_per_voxel = int(max_points_per_voxel)
        cfg.max_occ_voxels = int(max_occ_voxels_per_example)
        cfg.ranges[:] = self.ranges
        self.compat = tuple(compat)
        bad = set(self.compat) - {"truncate", "layered"}
        if bad:
            raise ValueError(f"VoxelGrid: unknown compat switches {sorted(bad)}")
        cfg.compat = (KNN_TRUNCATE if "truncate" in self.compat else 0) | (KNN_LAYERED if "layered" in self.compat else 0)
        self.max_points_per_voxel, self.max_occ_voxels = int(max_points_per_voxel), int(max_occ_voxels_per_example)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().spf_grid_create(C.byref(cfg), C.byref(self._h)), "spf_grid_create")
        self._built_for = None
        self._points = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().spf_grid_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ reference API
    def set_pointset(self, points, actual_num_points=None):
        """points: float32 [1,N,3] (or [N,3]) on the GPU; actual_num_points: int32 [1] (optional).
        The reference calls this 3-4 times per step on an unchanged buffer; rebuilding is skipped
        when the tensor (storage, version counter, length) is the one already built."""
        pts = points.reshape(-1, 3)
        if pts.dtype != torch.float32 or not pts.is_cuda:
            raise TypeError("VoxelGrid.set_pointset expects a float32 CUDA tensor")
        n = pts.shape[0]
        if actual_num_points is not None and not isinstance(actual_num_points, torch.Tensor):
            n = int(actual_num_points)
        key = (pts.data_ptr(), pts._version, n, pts.device.index)
        if key == self._built_for:
            return
        if actual_num_points is not None and isinstance(actual_num_points, torch.Tensor):
            n = min(n, int(actual_num_points.reshape(-1)[0].item()))
            key = key[:2] + (n,) + key[3:]
        pts = pts.contiguous()
        with torch.cuda.device(pts.device):
            _lib.check(_lib.lib().spf_grid_build(self._h, _lib.ptr(pts), n, _lib.stream_ptr()), "spf_grid_build")
        self._built_for = key
        self._points = pts  # keep the cloud alive: later kernels gather from it
        if "truncate" not in self.compat:      # the exact specification keeps everything: say where upstream's capacity limits would bite
            gi = self.info()
            if gi["max_cell_points"] > self.max_points_per_voxel > 0 or gi["n_occupied"] > self.max_occ_voxels > 0:
                import warnings

                warnings.warn(f"VoxelGrid: fullest cell holds {gi['max_cell_points']} points (max_points_per_voxel = {self.max_points_per_voxel}), "
                              f"{gi['n_occupied']} cells are occupied (max_occ_voxels = {self.max_occ_voxels}): upstream torch_knnquery would drop "
                              "points / cells here, at random; this build keeps all of them (compat=('truncate',) applies the limits "
                              "deterministically)", stacklevel=2)

    def query(self, raypos, k, radius_limit_scale, max_shading_points_per_ray):
        """raypos float32 [1,R,D,3] -> (sample_pidx int32 [1,Rv,SR,k] (-1 pad),
        sample_loc float32 [1,Rv,SR,3], ray_mask int8 [1,R]).  Rv is data dependent, so this
        compatibility form synchronises; the model's own path uses query_dense instead."""
        if raypos.dim() != 4 or raypos.shape[0] != 1:
            raise ValueError("VoxelGrid.query expects raypos of shape [1,R,D,3]")
        d = self.query_dense(raypos[0], k, radius_limit_scale, max_shading_points_per_ray)
        keep = d["ray_valid"].bool()
        return d["pidx"][keep].unsqueeze(0), d["loc"][keep].unsqueeze(0), d["ray_valid"].to(torch.int8).unsqueeze(0)

    # ------------------------------------------------------------------ native (no-sync) form
    def query_dense(self, raypos, k, radius_limit_scale, max_shading_points_per_ray, slots=None):
        """raypos float32 [R,D,3] -> dict of worst-case-sized device tensors (no host sync):
        pidx [R,SR,k] i32, loc [R,SR,3] f32, slot_sample [R,SR] i32, slot_valid [R,SR] u8, ray_valid [R] u8.
        slots: (slot_sample int32 [R,SR], ray_valid uint8 [R] cleared) assigned by the caller (spf_sampler_train): only the neighbour search runs."""
        if self._built_for is None:
            raise RuntimeError("VoxelGrid.query before set_pointset")
        x = raypos
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError("VoxelGrid.query expects a float32 CUDA tensor")
        x = x.contiguous()
        R, D = int(x.shape[0]), int(x.shape[1])
        SR, k = int(max_shading_points_per_ray), int(k)
        dev = x.device
        out = {
            "pidx": torch.empty((R, SR, k), dtype=torch.int32, device=dev),
            "loc": torch.empty((R, SR, 3), dtype=torch.float32, device=dev),
            "slot_sample": torch.empty((R, SR), dtype=torch.int32, device=dev) if slots is None else slots[0],
            "slot_valid": torch.empty((R, SR), dtype=torch.uint8, device=dev),
            "ray_valid": torch.empty((R,), dtype=torch.uint8, device=dev) if slots is None else slots[1],
        }
        with torch.cuda.device(dev), _prof.span("knn", rays=R, samples_per_ray=D, slots=SR, k=k, hit_slots=out["slot_valid"]):
            if slots is None:
                _lib.check(_lib.lib().spf_grid_query(self._h, _lib.ptr(x), R, D, k, float(radius_limit_scale), SR,
                                                     _lib.ptr(out["pidx"]), _lib.ptr(out["loc"]), _lib.ptr(out["slot_sample"]),
                                                     _lib.ptr(out["slot_valid"]), _lib.ptr(out["ray_valid"]), _lib.stream_ptr()),
                           "spf_grid_query")
            else:
                if tuple(slots[0].shape) != (R, SR) or slots[0].dtype != torch.int32 or tuple(slots[1].shape) != (R,):
                    raise ValueError("query_dense: slots = (int32 [R,SR], uint8 [R])")
                _lib.check(_lib.lib().spf_grid_knn(self._h, _lib.ptr(x), R, D, k, float(radius_limit_scale), SR, _lib.ptr(slots[0]),
                                                   _lib.ptr(out["pidx"]), _lib.ptr(out["loc"]), _lib.ptr(out["slot_valid"]), _lib.ptr(out["ray_valid"]),
                                                   _lib.stream_ptr()), "spf_grid_knn")
        return out

    def sweep_hits(self, xs, ys, zs, fill, first, count, pts, idx, counter, fill_value=1000.0):
        """spf_grid_sweep_hits: for the flat indices [first, first + count) of np.meshgrid(xs, ys, zs) ('xy', raveled) write `fill_value` into
        fill[: count] where the point fails the dilated-occupancy test and append the others to (pts [*,3], idx int64 [*]) at counter[0]."""
        if self._built_for is None:
            raise RuntimeError("VoxelGrid.sweep_hits before set_pointset")
        with torch.cuda.device(fill.device):
            _lib.check(_lib.lib().spf_grid_sweep_hits(self._h, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(zs), xs.numel(), ys.numel(), zs.numel(), int(first), int(count),
                                                      _lib.ptr(fill), float(fill_value), _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(counter), _lib.stream_ptr()),
                       "spf_grid_sweep_hits")

    def info(self):
        gi = _lib.GridInfo()
        _lib.check(_lib.lib().spf_grid_get_info(self._h, C.byref(gi)), "spf_grid_get_info")
        return {"origin": tuple(gi.origin), "cell": tuple(gi.cell), "dims": tuple(gi.dims), "n_points": gi.n_points,
                "n_in_range": gi.n_in_range, "n_occupied": gi.n_occupied, "max_cell_points": gi.max_cell_points}
