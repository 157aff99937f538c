"""CPU: the drop-in registration exposes the reference's module / class / function names."""
import inspect
import sys


def test_reference_names_resolve_after_install():
    import spurfies_amd

    for name in [m for m in sys.modules if m == "torch_knnquery" or m.startswith("spurfies.") or m == "spurfies"]:
        del sys.modules[name]
    spurfies_amd.install_dropin()
    from torch_knnquery import VoxelGrid
    from spurfies.model.density import LaplaceDensity
    from spurfies.model.embedder import get_embedder
    from spurfies.model.loss import VolSDFLoss
    from spurfies.model.pointneus_disent import PointVolSDF
    from spurfies.model.ray_sampler import ErrorBoundSampler_pn, UniformSampler
    from spurfies.model.utils import get_keypoint_data, load_neural_points, mask_to_batch_ray_idx, query, query_geo, tv_regul
    from spurfies.utils.general import get_class
    from spurfies.utils.rend_util import get_camera_params, lift

    # signatures the reference's call sites rely on
    assert list(inspect.signature(VoxelGrid.__init__).parameters)[1:7] == [
        "voxel_size", "voxel_scale", "kernel_size", "max_points_per_voxel", "max_occ_voxels_per_example", "ranges"]
    # one optional extra after the reference's six: the upstream-compatibility switches, default = the frozen specification
    assert inspect.signature(VoxelGrid.__init__).parameters["compat"].default == ()
    assert list(inspect.signature(VoxelGrid.query).parameters)[1:] == ["raypos", "k", "radius_limit_scale", "max_shading_points_per_ray"]
    assert list(inspect.signature(PointVolSDF.__init__).parameters)[1:4] == ["conf", "scan_id", "dataset"]
    assert list(inspect.signature(PointVolSDF.forward).parameters)[1:] == ["input", "fast"]
    assert list(inspect.signature(ErrorBoundSampler_pn.get_z_vals).parameters)[1:] == ["ray_dirs", "cam_loc", "model", "fast", "iter_step"]
    assert list(inspect.signature(query).parameters) == ["voxel_grid", "inputs", "k", "r", "max_shading_pts"]
    assert get_class("spurfies.model.pointneus_disent.PointVolSDF") is PointVolSDF
    emb, dim = get_embedder(6, 3)
    assert dim == 39
    for m in ("get_sdf_eval", "sdf_importance", "pseudo_sdf", "volume_rendering", "get_importance_rays"):
        assert hasattr(PointVolSDF, m)


def test_ply_reader_and_voxel_thinning(tmp_path):
    """Load-time path (utils.py:59-88): binary PLY -> points + colours -> one point per occupied voxel."""
    import numpy as np
    import torch

    from spurfies_amd.model.utils import construct_vox_points_closest, load_neural_points

    rng = np.random.default_rng(0)
    n = 500
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, k in enumerate("xyz"):
        rec[k] = xyz[:, i]
    for i, k in enumerate(("red", "green", "blue")):
        rec[k] = rgb[:, i]
    path = tmp_path / "cloud.ply"
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
    path.write_bytes(header.encode() + rec.tobytes())
    full = load_neural_points(str(path), vox_res=None, device="cpu")
    assert np.array_equal(full["pts"].numpy(), xyz) and np.array_equal(full["colors"].numpy(), rgb)
    thin = load_neural_points(str(path), vox_res=4, device="cpu")
    pts = torch.from_numpy(xyz)
    _, grid_idx, idx = construct_vox_points_closest(pts, 4)
    assert len(thin["pts"]) == len(grid_idx) <= 64 and len(set(idx.tolist())) == len(idx)
    # the kept point of each voxel is the one closest to the voxel's centroid (scatter_min semantics)
    mn, mx = pts.min(0)[0], pts.max(0)[0]
    edge = (mx - mn).max() * 1.05
    cell = torch.floor((pts - ((mx + mn) / 2 - edge / 2)) / (edge / 4)).int()
    for v, keep in zip(grid_idx[:10], idx[:10]):
        members = (cell == v).all(-1).nonzero().flatten()
        cen = pts[members].mean(0)
        assert keep in members and torch.isclose((pts[keep] - cen).norm(), (pts[members] - cen).norm(dim=-1).min())
