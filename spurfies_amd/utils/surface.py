"""Surface-extraction front half of the reference's mesh route (spurfies/utils/plots.py:188-333): the
evaluation grid, the `get_sdf_eval` sweep over it (the biggest pure-inference consumer of the kNN + geometry kernels), the
zero-level surface points, and the symmetric Chamfer distance the acceptance criterion is stated in (evals/eval_dtu.py:120-254).

`surface_points` returns the points marching cubes places its vertices at — the linear-interpolation zero crossings along grid
edges — which is what Chamfer is measured on.  `triangulate` is this build's own iso-surface mesher (skimage's marching cubes is
absent here): marching TETRAHEDRA on the Kuhn split of every cell (six tetrahedra around the main diagonal; the split is
translation-invariant, so neighbouring cells agree on every shared face and the mesh is watertight wherever the volume is defined),
the same vertices on the cube edges plus vertices on face / body diagonals; `write_ply` stores it the way the reference exports
its meshes (plots.py:266-300 via trimesh)."""
from __future__ import annotations

import numpy as np
import torch

SDF_FILL = 1000.0


def get_grid(points, resolution, input_min=None, input_max=None, eps=0.1):
    """plots.py:302-333: cubic cells, `resolution` samples along the shortest axis of the bounding box, meshgrid in the
    reference's (x, y, z) 'xy' indexing -> {'grid_points' [M,3] float32 tensor (CPU), 'xyz': [x, y, z], ...}.
    Bit-for-bit the reference's arithmetic (tests/golden/mesh_grid.npz): the bounds keep the dtype they arrive in — float32 when they
    come from a point tensor, as torch.min(...).numpy() gives — so the shortest axis is a float32 linspace, the step a float32 scalar, and
    the two other axes float64 aranges of float32 scalars, exactly as numpy evaluates the reference's expressions."""
    if input_min is None or input_max is None:
        pts = torch.as_tensor(points)
        input_min, input_max = pts.min(0)[0].numpy(), pts.max(0)[0].numpy()
    input_min, input_max = np.asarray(input_min), np.asarray(input_max)
    s = int(np.argmin(input_max - input_min))
    lin = np.linspace(input_min[s] - eps, input_max[s] + eps, resolution)
    length = np.max(lin) - np.min(lin)
    step = length / (lin.shape[0] - 1)
    axes = []
    for a in range(3):
        axes.append(lin if a == s else np.arange(input_min[a] - eps, input_max[a] + step + eps, step))
    xx, yy, zz = np.meshgrid(*axes)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    return {"grid_points": grid_points, "shortest_axis_length": length, "xyz": axes, "shortest_axis_index": s}


def get_grid_uniform(resolution, grid_boundary=(-2.0, 2.0)):
    """plots.py:288-300: the same `resolution` samples of [lo, hi] on all three axes (the training-time plots' grid)."""
    x = np.linspace(grid_boundary[0], grid_boundary[1], resolution)
    xx, yy, zz = np.meshgrid(x, x, x)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    return {"grid_points": grid_points, "shortest_axis_length": 2.0, "xyz": [x, x, x], "shortest_axis_index": 0}


def sdf_volume(sdf, grid, splitn=100000, device="cuda"):
    """plots.py:249-253: `sdf` (e.g. model.get_sdf_eval) over the grid in `splitn`-point chunks -> float32 volume indexed
    [y, x, z] like np.meshgrid's 'xy' layout (1000 where a point has no neighbour).  The reference moves every chunk to the device and
    its result back (a synchronisation per chunk); here the grid goes over once, the chunks are evaluated back to back without a
    host round trip and the volume comes back in one copy."""
    x, y, zz = grid["xyz"]
    model = getattr(sdf, "__self__", None)
    if getattr(sdf, "__name__", "") == "get_sdf_eval" and hasattr(model, "sdf_eval_grid") and model.neural_pts.is_cuda:
        # the product model's own sweep: the same values (chunking is not observable), without the [M,3] upload, with the ~88 % of the grid
        # that fails the dilated-occupancy test filled at once and device-side chunks of 4 M points (PointVolSDF.sdf_eval_grid)
        return model.sdf_eval_grid(x, y, zz).cpu().numpy()
    with torch.no_grad():
        pts = grid["grid_points"].to(device)
        out = torch.empty((pts.shape[0],), dtype=torch.float32, device=device)
        for i in range(0, pts.shape[0], splitn):
            out[i: i + splitn] = sdf(pts[i: i + splitn]).detach().float()
    return out.cpu().numpy().reshape(len(y), len(x), len(zz))


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


# ---- own triangulation (marching tetrahedra) --------------------------------------------------------------------------------
# cube corner c = x + 2 y + 4 z (bits); Kuhn split: six tetrahedra that all contain the main diagonal 0-7
_TETS = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7], [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]], dtype=np.int64)


def _tet_table():
    """sign case (bit i = vertex i inside) -> list of triangles, each three edges (i, j) with i inside, j outside."""
    table = []
    for case in range(16):
        ins = [i for i in range(4) if case >> i & 1]
        outs = [i for i in range(4) if not case >> i & 1]
        if len(ins) in (0, 4):
            table.append([])
        elif len(ins) == 1:
            a = ins[0]
            table.append([[(a, outs[0]), (a, outs[1]), (a, outs[2])]])
        elif len(ins) == 3:
            d = outs[0]
            table.append([[(ins[0], d), (ins[1], d), (ins[2], d)]])
        else:
            a, b = ins
            c, d = outs
            table.append([[(a, c), (a, d), (b, d)], [(a, c), (b, d), (b, c)]])
    return table


_TET_TABLE = _tet_table()


def triangulate(volume, grid, level=0.0):
    """Iso-surface of `volume` ([y, x, z] as sdf_volume returns it) at `level` -> (vertices [V,3] float64, faces [F,3] int64),
    vertices welded (one per crossed grid edge / diagonal), faces wound so that normals point towards larger values (outside of a
    signed-distance volume).  Cells touching a no-neighbour sample (1000) are skipped."""
    x, y, z = (np.asarray(a, dtype=np.float64) for a in grid["xyz"])
    ny, nx, nz = volume.shape
    vol = volume.astype(np.float64) - level
    valid = volume != SDF_FILL
    # cells and their 8 corners (global vertex id = index into the raveled [y, x, z] volume)
    iy, ix, iz = np.meshgrid(np.arange(ny - 1), np.arange(nx - 1), np.arange(nz - 1), indexing="ij")
    corner_ids, corner_val, corner_ok = [], [], []
    for c in range(8):
        dx, dy, dz = c & 1, c >> 1 & 1, c >> 2 & 1
        gid = ((iy + dy) * nx + (ix + dx)) * nz + (iz + dz)
        corner_ids.append(gid.ravel())
    corner_ids = np.stack(corner_ids, 1)                                    # [cells, 8]
    flat_val, flat_ok = vol.ravel(), valid.ravel()
    cv = flat_val[corner_ids]
    active = flat_ok[corner_ids].all(1) & (cv.min(1) < 0) & (cv.max(1) >= 0)
    corner_ids, cv = corner_ids[active], cv[active]
    if corner_ids.shape[0] == 0:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    gy, gx, gz = np.unravel_index(np.arange(ny * nx * nz), (ny, nx, nz))

    def pos(ids):
        return np.stack([x[gx[ids]], y[gy[ids]], z[gz[ids]]], -1)

    keys, pts, tri_inside, tri_outside = [], [], [], []
    for tet in _TETS:
        tid, tv = corner_ids[:, tet], cv[:, tet]                            # [cells, 4]
        case = ((tv < 0) * (1 << np.arange(4))).sum(1)
        for cs in range(1, 15):
            sel = case == cs
            if not sel.any():
                continue
            ids, vals = tid[sel], tv[sel]
            ins = [i for i in range(4) if cs >> i & 1]
            outs = [i for i in range(4) if not cs >> i & 1]
            c_in, c_out = pos(ids[:, ins]).mean(1), pos(ids[:, outs]).mean(1)
            for tri in _TET_TABLE[cs]:
                for (i, j) in tri:
                    a, b = ids[:, i], ids[:, j]
                    t = vals[:, i] / (vals[:, i] - vals[:, j])
                    pa, pb = pos(a), pos(b)
                    pts.append(pa + t[:, None] * (pb - pa))
                    keys.append(np.stack([np.minimum(a, b), np.maximum(a, b)], 1))
                tri_inside.append(c_in)
                tri_outside.append(c_out)
    pts, keys = np.concatenate(pts, 0), np.concatenate(keys, 0)              # three consecutive blocks per triangle batch
    # rebuild per-triangle corner order: blocks were appended as [edge0 of batch, edge1 of batch, edge2 of batch] per batch
    sizes = [c.shape[0] for c in tri_inside]
    corners, off = [], 0
    for n in sizes:
        corners.append(np.stack([np.arange(off, off + n), np.arange(off + n, off + 2 * n), np.arange(off + 2 * n, off + 3 * n)], 1))
        off += 3 * n
    corners = np.concatenate(corners, 0)                                    # [F, 3] indices into pts / keys
    c_in, c_out = np.concatenate(tri_inside, 0), np.concatenate(tri_outside, 0)
    uniq, inverse = np.unique(keys, axis=0, return_inverse=True)
    inverse = inverse.reshape(-1)
    verts = np.zeros((uniq.shape[0], 3))
    verts[inverse] = pts                                                    # all copies of a welded vertex coincide
    faces = inverse[corners]
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    flip = (np.cross(p1 - p0, p2 - p0) * (c_out - c_in)).sum(1) < 0         # normal must point from the inside corners to the outside ones
    faces[flip] = faces[flip][:, [0, 2, 1]]
    good = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])   # a value exactly on a corner
    return verts, faces[good]


def write_ply(path, verts, faces):
    """Binary little-endian PLY (vertex x y z float32, face vertex_indices int32 lists)."""
    verts, faces = np.asarray(verts, dtype="<f4"), np.asarray(faces, dtype="<i4")
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\n"
                 f"property float z\nelement face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(verts.tobytes())
        rec = np.empty(len(faces), dtype=[("n", "u1"), ("v", "<i4", (3,))])
        rec["n"], rec["v"] = 3, faces
        f.write(rec.tobytes())


def write_ply_points(path, pts, colors=None):
    """Binary little-endian point-cloud PLY in the layout the reference's clouds use (vertex x y z float32 [+ red green
    blue uchar], dust3r_inference.py -> spurfies/model/utils.py:59-88)."""
    pts = np.asarray(pts, dtype="<f4")
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    header = f"ply\nformat binary_little_endian 1.0\nelement vertex {len(pts)}\nproperty float x\nproperty float y\nproperty float z\n"
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
    rec = np.empty(len(pts), dtype=fields)
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    if colors is not None:
        col = np.asarray(colors, dtype="u1")
        rec["red"], rec["green"], rec["blue"] = col[:, 0], col[:, 1], col[:, 2]
    with open(path, "wb") as f:
        f.write((header + "end_header\n").encode())
        f.write(rec.tobytes())


# -------------------------------------------------------------------------------------------------------------------------------
# Back half of the route: what the reference does with the mesh before it quotes a Chamfer number.
def largest_component(verts, faces):
    """plots.py:213-215: keep the connected component (faces linked through shared EDGES, as trimesh.split(only_watertight=False))
    with the largest surface area; vertices are re-indexed.  -> (verts, faces)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    verts, faces = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    if len(faces) == 0:
        return verts[:0], faces
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0), 1)
    fid = np.tile(np.arange(len(faces)), 3)
    order = np.lexsort((e[:, 1], e[:, 0]))
    e, fid = e[order], fid[order]
    same = np.all(e[1:] == e[:-1], axis=1)                      # consecutive entries share an edge
    a, b = fid[:-1][same], fid[1:][same]
    adj = coo_matrix((np.ones(len(a)), (a, b)), shape=(len(faces), len(faces)))
    _, label = connected_components(adj, directed=False)
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1)
    keep = label == np.argmax(np.bincount(label, weights=area))
    used = np.unique(faces[keep])
    remap = np.full(len(verts), -1, np.int64)
    remap[used] = np.arange(len(used))
    return verts[used], remap[faces[keep]]


def sample_mesh_points(verts, faces, thresh):
    """evals/eval_dtu.py:20-29,72-109 ('mesh' mode): the mesh vertices plus, per triangle, the lattice points
    (i + 0.5) / n1 v1 + (j + 0.5) / n2 v2 with i / n1 + j / n2 < 1 (n = floor(edge / (thresh sqrt(l1 l2 / 2A)))), so that the
    samples are about `thresh` apart on every triangle."""
    verts, faces = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    tri = verts[faces]
    v1, v2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    l1, l2 = np.linalg.norm(v1, axis=-1), np.linalg.norm(v2, axis=-1)
    area2 = np.linalg.norm(np.cross(v1, v2), axis=-1)
    ok = area2 > 0
    tri, v1, v2, l1, l2, area2 = tri[ok], v1[ok], v2[ok], l1[ok], l2[ok], area2[ok]
    thr = thresh * np.sqrt(l1 * l2 / area2)
    n1, n2 = np.floor(l1 / thr), np.floor(l2 / thr)
    # triangles with the same lattice share the barycentric offsets: evaluated per lattice, laid out in the REFERENCE's order (the vertices,
    # then triangle after triangle, each triangle's lattice points row-major) — the order matters downstream: the evaluation shuffles the
    # samples with a seeded permutation and thins them greedily (downsample_points)
    keys, inv = np.unique(np.stack([n1, n2], 1), axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    lattices = []
    for key in keys:
        c = np.mgrid[: int(key[0]) + 1, : int(key[1]) + 1].astype(np.float64) + 0.5
        c[0] /= max(key[0], 1e-7)
        c[1] /= max(key[1], 1e-7)
        kk = c.reshape(2, -1).T
        lattices.append(kk[kk.sum(-1) < 1])
    per_tri = np.asarray([len(lattices[g]) for g in range(len(keys))], np.int64)[inv]
    start = np.concatenate([[0], np.cumsum(per_tri)])
    new_pts = np.empty((int(start[-1]), 3), np.float64)
    for g, kk in enumerate(lattices):
        sel = np.nonzero(inv == g)[0]
        if len(kk) and len(sel):
            q = v1[sel][:, None, :] * kk[None, :, :1] + v2[sel][:, None, :] * kk[None, :, 1:] + tri[sel][:, :1, :]
            new_pts[(start[sel][:, None] + np.arange(len(kk))[None, :]).reshape(-1)] = q.reshape(-1, 3)
    return np.concatenate([verts, new_pts], 0)


def downsample_points(points, thresh, seed=None):
    """evals/eval_dtu.py:118-138: shuffle, then greedily keep a point and strike out everything within `thresh` of it."""
    from scipy.spatial import cKDTree

    pts = np.array(points, np.float64)
    np.random.default_rng(seed).shuffle(pts, axis=0)
    nbrs = cKDTree(pts).query_ball_point(pts, thresh)
    mask = np.ones(len(pts), bool)
    for cur, idxs in enumerate(nbrs):
        if mask[cur]:
            mask[idxs] = False
            mask[cur] = True
    return pts[mask]


def chamfer_dtu(data_pts, gt_pts, max_dist=20.0, thresh=None, seed=0, bbox=None, obs_mask=None, ground_plane=None, patch=60.0):
    """evals/eval_dtu.py:118-254, pinned to the script itself by tests/golden/eval_dtu.npz: optional greedy down-sampling of the
    reconstruction at `thresh` (:118-138), then
      * `obs_mask` = {'ObsMask' bool [X,Y,Z], 'BB' [2,3], 'Res'} (the scan's ObsMask*.mat): points inside BB widened by `patch` below and
        2 x `patch` above are `data_in` (:146-149); of those, the ones whose voxel round((p - BB[0]) / Res) lies in the grid and is observed
        are `data_in_obs` (:151-159) — accuracy = mean distance data_in_obs -> ground truth over the distances below `max_dist` (:170-176);
      * `ground_plane` P [4]: ground-truth points with P . (x, 1) > 0 (:200-202) — completeness = mean distance of those to **data_in**
        (the in-bound set, NOT the observed subset and not the whole cloud, :204-208), again below `max_dist`;
      * without `obs_mask`, an axis-aligned `bbox` = (lo [3], hi [3]) plays the role of the in-bound test and data_in_obs = data_in.
    Units are those of the inputs (the reference works in millimetres: thresh 0.2, max_dist 20).
    -> dict(accuracy, completeness, overall, n_data (= |data_in_obs|), n_down, n_gt)."""
    from scipy.spatial import cKDTree

    data = np.asarray(data_pts, np.float64)
    gt = np.asarray(gt_pts, np.float64)
    if thresh is not None:
        data = downsample_points(data, thresh, seed)
    data_in = data_obs = data
    if obs_mask is not None:
        mask, bb, res = np.asarray(obs_mask["ObsMask"]), np.asarray(obs_mask["BB"], np.float32), obs_mask["Res"]
        inbound = ((data >= bb[:1] - patch) & (data < bb[1:] + patch * 2)).sum(axis=-1) == 3
        data_in = data[inbound]
        cell = np.around((data_in - bb[:1]) / res).astype(np.int32)
        inside = ((cell >= 0) & (cell < np.expand_dims(mask.shape, 0))).sum(axis=-1) == 3
        cell_in = cell[inside]
        data_obs = data_in[inside][mask[cell_in[:, 0], cell_in[:, 1], cell_in[:, 2]].astype(bool)]
    elif bbox is not None:
        lo, hi = np.asarray(bbox[0], np.float64), np.asarray(bbox[1], np.float64)
        data_in = data_obs = data[np.all((data >= lo) & (data < hi), axis=1)]
    gt_above = gt
    if ground_plane is not None:
        hom = np.concatenate([gt, np.ones_like(gt[:, :1])], -1)
        gt_above = gt[(np.asarray(ground_plane, np.float64).reshape(1, 4) * hom).sum(-1) > 0]
    d2s = cKDTree(gt).query(data_obs)[0]
    s2d = cKDTree(data_in).query(gt_above)[0]
    acc = float(d2s[d2s < max_dist].mean()) if (d2s < max_dist).any() else float("nan")
    comp = float(s2d[s2d < max_dist].mean()) if (s2d < max_dist).any() else float("nan")
    return {"accuracy": acc, "completeness": comp, "overall": 0.5 * (acc + comp), "n_data": len(data_obs), "n_down": len(data), "n_gt": len(gt_above)}



def extract_surface(sdf, resolution, input_min, input_max, splitn=100000, device="cuda", keep_largest=True):
    """get_surface_by_grid's plain branch (plots.py:188-287 with higher_res=False): reference-shaped grid over the box, chunked SDF
    sweep, iso-surface at 0, largest component.  -> (verts, faces, volume, grid); (None, None, volume, grid) when the SDF does not
    cross zero inside the box."""
    grid = get_grid(None, resolution, input_min=np.asarray(input_min), input_max=np.asarray(input_max), eps=0.0)
    vol = sdf_volume(sdf, grid, splitn=splitn, device=device)
    ok = vol != SDF_FILL
    if not ok.any() or vol[ok].min() > 0 or vol[ok].max() < 0:
        return None, None, vol, grid
    verts, faces = triangulate(vol, grid)
    if keep_largest and len(faces):
        verts, faces = largest_component(verts, faces)
    return verts, faces, vol, grid
