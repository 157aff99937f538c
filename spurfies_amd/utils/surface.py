"""Surface-extraction front half of the reference's mesh route (spurfies/utils/plots.py:188-333): the
evaluation grid, the `get_sdf_eval` sweep over it (the biggest pure-inference consumer of the kNN + geometry kernels), the
zero-level surface points, and the symmetric Chamfer distance the acceptance criterion is stated in (evals/eval_dtu.py:120-254).

Marching-cubes triangulation itself (skimage, absent here) is not reproduced: `surface_points` returns the points marching
cubes places its vertices at — the linear-interpolation zero crossings along grid edges — which is what Chamfer is measured on."""
from __future__ import annotations

import numpy as np
import torch

SDF_FILL = 1000.0


def get_grid(points, resolution, input_min=None, input_max=None, eps=0.1):
    """plots.py:302-333: cubic cells, `resolution` samples along the shortest axis of the bounding box, meshgrid in the
    reference's (x, y, z) 'xy' indexing -> {'grid_points' [M,3] float32 tensor (CPU), 'xyz': [x, y, z], ...}."""
    if input_min is None or input_max is None:
        pts = torch.as_tensor(points)
        input_min, input_max = pts.min(0)[0].numpy(), pts.max(0)[0].numpy()
    input_min, input_max = np.asarray(input_min, dtype=np.float64), np.asarray(input_max, dtype=np.float64)
    s = int(np.argmin(input_max - input_min))
    lin = np.linspace(input_min[s] - eps, input_max[s] + eps, resolution)
    length = lin.max() - lin.min()
    step = length / (lin.shape[0] - 1)
    axes = []
    for a in range(3):
        axes.append(lin if a == s else np.arange(input_min[a] - eps, input_max[a] + step + eps, step))
    xx, yy, zz = np.meshgrid(*axes)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    return {"grid_points": grid_points, "shortest_axis_length": length, "xyz": axes, "shortest_axis_index": s}


def sdf_volume(sdf, grid, splitn=100000, device="cuda"):
    """plots.py:249-253: `sdf` (e.g. model.get_sdf_eval) over the grid in `splitn`-point chunks -> float32 volume indexed
    [y, x, z] like np.meshgrid's 'xy' layout (1000 where a point has no neighbour)."""
    z = []
    with torch.no_grad():
        for pnts in torch.split(grid["grid_points"], splitn, dim=0):
            z.append(sdf(pnts.to(device)).detach().float().cpu().numpy())
    x, y, zz = grid["xyz"]
    return np.concatenate(z, 0).astype(np.float32).reshape(len(y), len(x), len(zz))


def surface_points(volume, grid, level=0.0):
    """Zero crossings of `volume - level` along the grid edges (where marching cubes puts its vertices); edges touching a
    no-neighbour sample (1000) are skipped."""
    x, y, z = grid["xyz"]
    gx, gy, gz = np.meshgrid(x, y, z)
    coords = np.stack([gx, gy, gz], -1)
    vol = volume.astype(np.float64) - level
    valid = volume != SDF_FILL
    pts = []
    for ax in range(3):
        s0, s1 = [slice(None)] * 3, [slice(None)] * 3
        s0[ax], s1[ax] = slice(0, vol.shape[ax] - 1), slice(1, vol.shape[ax])
        a, c = vol[tuple(s0)], vol[tuple(s1)]
        ok = valid[tuple(s0)] & valid[tuple(s1)] & (a * c < 0)
        t = (a / (a - c + 1e-300))[ok]
        p0, p1 = coords[tuple(s0)][ok], coords[tuple(s1)][ok]
        pts.append(p0 + t[:, None] * (p1 - p0))
    return np.concatenate(pts, 0) if pts else np.zeros((0, 3))


def chamfer(a, b):
    """Symmetric Chamfer distance (mean nearest-neighbour distance both ways, averaged), the DTU evaluation's accuracy /
    completeness pair collapsed to one number (evals/eval_dtu.py:238-254)."""
    from scipy.spatial import cKDTree

    da, _ = cKDTree(b).query(a)
    db, _ = cKDTree(a).query(b)
    return 0.5 * (float(da.mean()) + float(db.mean())), float(da.mean()), float(db.mean())
