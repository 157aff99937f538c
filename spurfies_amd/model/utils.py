"""Glue around the native kNN op with the reference's function names
(spurfies/model/utils.py:59-183, 221-282).

`query`, `query_geo`, `get_keypoint_data`, `mask_to_batch_ray_idx` keep the reference's
(data-dependent-shape, host-synchronising) return conventions for code written against them;
PointVolSDF itself uses the dense, sync-free forms (VoxelGrid.query_dense + spf_compact_points).
"""
from __future__ import annotations

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# reference-convention wrappers
def query(voxel_grid, inputs, k, r, max_shading_pts):
    """utils.py:90-113.  inputs [R,D,3] -> (neighbor_idx int64 [P,k] (-1 pad), shading_pts [P,3],
    mask bool [R,SR], ray_mask bool [R])."""
    num_rays = inputs.shape[0]
    d = voxel_grid.query_dense(inputs, k, r, max_shading_pts)
    mask = d["slot_valid"].bool()
    neighbor_idx = d["pidx"][mask].to(torch.int64)
    shading_pts = d["loc"][mask]
    return neighbor_idx, shading_pts, mask.view(num_rays, -1), d["ray_valid"].bool()


def query_geo(voxel_grid, inputs, k, r):
    """utils.py:115-138 (the SR = 1 form)."""
    return query(voxel_grid, inputs, k, r, 1)


def mask_to_batch_ray_idx(valid_neighbor_mask):
    """utils.py:172-183: row id of every valid (point, neighbour) pair."""
    n = valid_neighbor_mask.shape[0]
    rows = torch.arange(n, device=valid_neighbor_mask.device).view(-1, 1)
    return torch.masked_select(rows, valid_neighbor_mask)


def get_keypoint_data(neighbor_idx, mask, kp_pos=None, kp_feat=None, kp_geometry=None):
    """utils.py:140-170 without the per-call concat of the whole table: index each table directly."""
    flat = neighbor_idx[mask]
    res = {}
    if kp_pos is not None:
        res["pos"] = kp_pos[flat]
    if kp_feat is not None:
        res["feat"] = kp_feat[flat]
    if kp_geometry is not None:
        res["feat_geometry"] = kp_geometry[flat]
    return res


# ---------------------------------------------------------------------------------------------
# total-variation regulariser over the (static) neighbour graph of the cloud
class TVGraph:
    """Neighbour graph of utils.py:221-282, built ONCE: the cloud is a constant buffer, so the
    kNN among neural points, the self-removal rule and the inverse-distance weights never change."""

    def __init__(self, voxel_grid, kp_pos, k, r):
        n = kp_pos.shape[0]
        dev = kp_pos.device
        d = voxel_grid.query_dense(kp_pos.detach().view(n, 1, 3), k, r, 1)
        nb = d["pidx"].view(n, k).to(torch.int64)
        valid_kp = d["slot_valid"].view(n).bool()
        own = torch.arange(n, device=dev)
        padded = torch.full((n, k), -1, dtype=torch.int64, device=dev)
        padded[:, 0] = own                                   # points the grid lost keep themselves
        padded = torch.where(valid_kp[:, None], nb, padded)
        ident = padded == own[:, None]
        enough = (padded >= 0).sum(-1, keepdim=True) > 1
        padded = torch.where(ident & enough, torch.full_like(padded, -1), padded)
        self.valid = padded >= 0                              # [n,k]
        nbr = padded.clamp(min=0)
        dist = torch.linalg.norm(kp_pos[nbr] - kp_pos[:, None, :], dim=-1)
        w = 1 / (dist + 1.0e-5)
        self.w = torch.where(self.valid, w, torch.zeros_like(w)).contiguous()
        self.norm = self.w.sum(-1).contiguous()
        self.nbr = nbr.to(torch.int32).contiguous()

    def loss(self, kp_feat, reduce=True):
        """reduce=False: the per-point terms [n] instead of their mean (for the fused loss kernels, which form the mean themselves)."""
        from .. import ops

        return ops.TVLoss.apply(kp_feat, self.nbr, self.w, self.norm, reduce)


def tv_regul(voxel_grid, kp_pos, kp_feat, k, r):
    """utils.py:221-282, reference signature (graph rebuilt per call; PointVolSDF caches it)."""
    return TVGraph(voxel_grid, kp_pos, k, r).loss(kp_feat)


# ---------------------------------------------------------------------------------------------
# load-time: .ply cloud -> voxel-thinned neural points (utils.py:6-88, replaces plyfile + torch_scatter)
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def read_ply_vertices(path):
    """Minimal PLY reader (ascii / binary_little_endian / binary_big_endian), vertex element only."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return {name: data[:, i] for i, (name, _) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, order + t) for name, t in props])
        arr = np.frombuffer(f.read(n_vert * dt.itemsize), dtype=dt, count=n_vert)
        return {name: arr[name] for name, _ in props}


def construct_vox_points_closest(xyz, vox_res):
    """utils.py:6-36: per occupied voxel of a vox_res^3 grid over the 1.05x bounding cube, the index
    of the point closest to the voxel's centroid."""
    # the three scalars of the grid are formed on the host in float32, op for op as torch-CPU runs the reference's lines 10-21
    # (on a device `edge / vox_res` with a Python-scalar divisor becomes a multiplication by the reciprocal: one ulp off)
    xyz_min, xyz_max = xyz.min(dim=-2)[0].cpu(), xyz.max(dim=-2)[0].cpu()
    edge = torch.max(xyz_max - xyz_min) * 1.05
    space_min = (xyz_max + xyz_min) / 2 - edge / 2
    vsz = edge / vox_res
    if xyz.is_cuda:         # spf_voxel_cells: the same float32 subtraction / IEEE division / floor the CPU performs
        import ctypes as C

        from .. import _lib

        xyz_c = xyz.detach().float().contiguous()
        cell = torch.empty((xyz_c.shape[0], 3), dtype=torch.int32, device=xyz.device)
        mn = (C.c_float * 3)(*[float(v) for v in space_min.float().tolist()])
        with torch.cuda.device(xyz.device):
            _lib.check(_lib.lib().spf_voxel_cells(_lib.ptr(xyz_c), xyz_c.shape[0], C.byref(mn), float(vsz.float().item()), _lib.ptr(cell),
                                                  _lib.stream_ptr()), "spf_voxel_cells")
    else:
        cell = torch.floor((xyz - space_min[None]) / vsz).to(torch.int32)
    grid_idx, inv = torch.unique(cell, dim=0, return_inverse=True)
    m = grid_idx.shape[0]
    cnt = torch.zeros(m, device=xyz.device).index_add_(0, inv, torch.ones(len(xyz), device=xyz.device))
    centroid = torch.zeros(m, 3, device=xyz.device).index_add_(0, inv, xyz) / cnt[:, None]
    resid = torch.norm(xyz - centroid[inv], dim=-1)
    best = torch.full((m,), float("inf"), device=xyz.device).scatter_reduce(0, inv, resid, reduce="amin")
    is_best = resid == best[inv]
    ids = torch.arange(len(xyz), device=xyz.device)
    min_idx = torch.full((m,), len(xyz), dtype=torch.int64, device=xyz.device).scatter_reduce(
        0, inv[is_best], ids[is_best], reduce="amin")
    return centroid, grid_idx, min_idx


def voxelize(pointcloud, vox_res):
    _, _, idx = construct_vox_points_closest(pointcloud, vox_res)
    return pointcloud[idx], idx


def load_neural_points(path, vox_res=None, device="cuda"):
    """utils.py:59-88: {'pts': [N,3] float, 'colors': [N,3] 0..255 (if the file has them)}."""
    v = read_ply_vertices(path)
    pts = torch.from_numpy(np.stack([v["x"], v["y"], v["z"]], -1).astype(np.float32)).to(device)
    idx = None
    if vox_res is not None:
        pts, idx = voxelize(pts, vox_res)
    out = {"pts": pts}
    if "red" in v:
        col = torch.from_numpy(np.stack([v["red"], v["green"], v["blue"]], -1)).to(device)
        out["colors"] = col[idx] if idx is not None else col
    return out
