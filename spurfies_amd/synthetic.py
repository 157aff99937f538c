"""Synthetic DTU-shaped scenes.

The reference ships neither data nor checkpoints (readme.md:38-50), so every test and
benchmark input is generated here, deterministically, with numpy only:

* a neural point cloud thinned to ~`spacing` on a closed analytic surface (a lobed sphere),
* per-point colours in 0..255 (as `load_neural_points` would return, model/utils.py:59-88),
* DTU-like pinhole cameras (datasets/dtu.py:113-118 scales K to 576x768) on a ring,
* Kaiming-uniform MLP weights in the reference's `state_dict` key layout
  (pointneus_disent.py:76-107).

Nothing in this module touches the GPU or the oracle.
"""
from __future__ import annotations

import math
import numpy as np

IMG_H, IMG_W = 576, 768
FX = FY = 1388.0
CX, CY = 384.0, 288.0


def lobed_radius(dirs: np.ndarray, base: float) -> np.ndarray:
    """Radius of the analytic surface along unit directions `dirs` [n,3]."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    theta = np.arccos(np.clip(z, -1.0, 1.0))
    phi = np.arctan2(y, x)
    return base * (1.0 + 0.18 * np.sin(3.0 * theta) * np.cos(2.0 * phi) + 0.06 * np.cos(5.0 * phi) * np.sin(theta) ** 2)


def make_cloud(n_target: int = 10000, spacing: float = 0.025, seed: int = 0, jitter: float = 0.15):
    """Quasi-uniform cloud on the lobed sphere whose base radius is chosen so that a
    Fibonacci lattice of `n_target` points has ~`spacing` nearest-neighbour distance.

    Returns (pts float32 [N,3], colors float32 [N,3] in 0..255, base_radius).
    """
    rng = np.random.default_rng(seed)
    base = math.sqrt(n_target * spacing * spacing / (4.0 * math.pi))
    i = np.arange(n_target, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * i / n_target
    rad = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    golden = math.pi * (3.0 - math.sqrt(5.0))
    ang = golden * i
    dirs = np.stack([rad * np.cos(ang), rad * np.sin(ang), z], -1)
    dirs = dirs + jitter * spacing / base * rng.standard_normal(dirs.shape)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    r = lobed_radius(dirs, base)
    pts = (dirs * r[:, None]).astype(np.float32)
    colors = np.floor(rng.uniform(0.0, 256.0, size=(n_target, 3))).clip(0, 255).astype(np.float32)
    return pts, colors, base


def look_at_pose(cam_pos: np.ndarray, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """Camera-to-world 4x4 (OpenCV convention: +z forward, +y down) as the reference's
    `pose` input (rend_util.py:60-95 uses pose[:3,:3] @ x_cam + pose[:3,3])."""
    cam_pos = np.asarray(cam_pos, np.float64)
    fwd = np.asarray(target, np.float64) - cam_pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, cam_pos
    return pose.astype(np.float32)


def make_cameras(n_views: int = 3, ring_radius: float = 2.2, height: float = 0.4):
    """(intrinsics [4,4] f32, poses [n_views,4,4] f32)."""
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = FX, FY, CX, CY
    poses = []
    for v in range(n_views):
        a = 2.0 * math.pi * v / max(n_views, 1) * 0.25 + 0.3  # a 90-degree arc, like 3 nearby DTU views
        poses.append(look_at_pose([ring_radius * math.cos(a), ring_radius * math.sin(a), height]))
    return K, np.stack(poses)


def make_pixels(n_rays: int, generator) -> "np.ndarray":
    """Pixel coordinates drawn the way datasets/dtu.py:360-364 does: randperm(H*W)[:n].
    `generator` is a torch.Generator (CPU); returns float32 [n,2] as (x, y)."""
    import torch

    perm = torch.randperm(IMG_H * IMG_W, generator=generator)[:n_rays]
    ys = torch.div(perm, IMG_W, rounding_mode="floor").float()
    xs = (perm % IMG_W).float()
    return torch.stack([xs, ys], -1).numpy()


_MLP_SHAPES = {
    # name -> list of (out, in); indices follow nn.Sequential positions in the reference
    "F_color": [(0, 256, 103), (2, 256, 256), (4, 256, 256), (6, 256, 256)],
    "F_geometry": [(0, 256, 35), (2, 256, 256), (4, 256, 256), (6, 256, 256), (8, 256, 256)],
    "T": [(0, 1, 256)],
    "R": [(0, 256, 277), (2, 256, 256), (4, 3, 256)],
}


def make_mlp_weights(seed: int = 0, geo_gain: float = 1.0) -> dict:
    """Kaiming-uniform weights/biases (nn.Linear's default init law) keyed like the
    reference state_dict: 'F_geometry.0.weight', ..., 'T.0.bias', 'R.4.weight'."""
    rng = np.random.default_rng(seed + 1234)
    out = {}
    for name, layers in _MLP_SHAPES.items():
        for pos, fo, fi in layers:
            bound = 1.0 / math.sqrt(fi)
            g = geo_gain if name in ("F_geometry", "T") else 1.0
            out[f"{name}.{pos}.weight"] = (g * rng.uniform(-bound, bound, size=(fo, fi))).astype(np.float32)
            out[f"{name}.{pos}.bias"] = (g * rng.uniform(-bound, bound, size=(fo,))).astype(np.float32)
    return out


def make_latents(n: int, colors: np.ndarray, seed: int = 0, geo_std: float = 0.01, feat_dim: int = 64):
    """Latent tables initialised as pointneus_disent.py:116-129,185-199 does:
    colour U(-1e-4,1e-4) with RGB*2/255-1 in dims 0..2; geometry N(0,std) with rows
    rescaled to norm<=1."""
    rng = np.random.default_rng(seed + 99)
    fc = rng.uniform(-1e-4, 1e-4, size=(n, feat_dim)).astype(np.float32)
    fc[:, :3] = colors * np.float32(2.0) / np.float32(255.0) - np.float32(1.0)
    fg = (geo_std * rng.standard_normal((n, feat_dim // 2))).astype(np.float32)
    norms = np.linalg.norm(fg, axis=-1, keepdims=True)
    fg = (fg * (np.minimum(norms, 1.0) / (norms + 1e-7))).astype(np.float32)
    return fc, fg


def surface_normals(pts: np.ndarray, base: float) -> np.ndarray:
    """Outward unit normals of the lobed sphere at (points near) its surface: gradient of f(p) = |p| - R(p / |p|) by
    central differences in float64."""
    p = pts.astype(np.float64)

    def f(q):
        r = np.linalg.norm(q, axis=-1)
        return r - lobed_radius(q / r[:, None], base)

    h = 1e-5
    g = np.stack([(f(p + h * e) - f(p - h * e)) / (2 * h) for e in np.eye(3)], -1)
    return (g / np.linalg.norm(g, axis=-1, keepdims=True)).astype(np.float32)


def surface_sdf(x: np.ndarray, base: float) -> np.ndarray:
    """First-order signed distance to the lobed sphere: f / |grad f| with f(p) = |p| - R(p / |p|)."""
    p = x.astype(np.float64)

    def f(q):
        r = np.linalg.norm(q, axis=-1)
        return r - lobed_radius(q / r[:, None], base)

    h = 1e-5
    g = np.stack([(f(p + h * e) - f(p - h * e)) / (2 * h) for e in np.eye(3)], -1)
    return f(p) / np.linalg.norm(g, axis=-1)


PRIOR_FITTED = None


def fitted_prior() -> dict:
    """F_geometry / T weights fitted by tools/fit_prior.py (the reference's ckpt/local_prior.pt is a separate download):
    with geometry latents g[:3] = 0.5 n they turn an oriented cloud into its signed-distance field."""
    global PRIOR_FITTED
    if PRIOR_FITTED is None:
        import os

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "prior_fitted.npz")
        PRIOR_FITTED = {k: v for k, v in np.load(path).items()}
    return PRIOR_FITTED


def make_scene(n_points: int = 10000, seed: int = 0, spacing: float = 0.025, geo_std: float = 0.3, prior: str = "kaiming") -> dict:
    """Everything a step needs, as numpy arrays (state_dict-style keys).
    prior='kaiming': random F_geometry / T and random geometry latents (the SDF is a generic smooth function);
    prior='fitted': the fitted local prior + latents carrying the analytic normals, i.e. the SDF is the signed distance
    to the analytic surface (realistic sampler convergence, zero crossings, pair counts)."""
    pts, colors, base = make_cloud(n_points, spacing, seed)
    K, poses = make_cameras()
    sd = make_mlp_weights(seed)
    if prior == "fitted":
        fc, _ = make_latents(len(pts), colors, seed, geo_std)
        rng = np.random.default_rng(seed + 77)
        fg = (0.05 * rng.standard_normal((len(pts), 32))).astype(np.float32)
        fg[:, :3] = np.float32(0.5) * surface_normals(pts, base)
        sd.update(fitted_prior())
    elif prior == "kaiming":
        fc, fg = make_latents(len(pts), colors, seed, geo_std)
    else:
        raise ValueError(prior)
    sd["neural_pts"] = pts
    sd["neural_feats_color"] = fc
    sd["neural_feats_geometry"] = fg
    sd["density.beta"] = np.float32(0.1)
    return {"state": sd, "colors": colors, "intrinsics": K, "poses": poses, "base_radius": base, "prior": prior,
            "ranges": (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0) if base * 1.3 < 1.0 else (-2.0, -2.0, -2.0, 2.0, 2.0, 2.0)}


def _smooth_field(rng, channels: int, h: int, w: int, coarse=(9, 12)) -> np.ndarray:
    """[channels, h, w] float32: coarse Gaussian noise, separably linearly interpolated (numpy only, deterministic)."""
    ch, cw = coarse
    base = rng.standard_normal((channels, ch, cw))
    ys, xs = np.linspace(0.0, ch - 1.0, h), np.linspace(0.0, cw - 1.0, w)
    y0, x0 = np.minimum(ys.astype(np.int64), ch - 2), np.minimum(xs.astype(np.int64), cw - 2)
    fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
    rows = base[:, y0, :] * (1.0 - fy) + base[:, y0 + 1, :] * fy               # [C, h, cw]
    return (rows[:, :, x0] * (1.0 - fx) + rows[:, :, x0 + 1] * fx).astype(np.float32)


def make_local_data(scene: dict, view: int, channels: int = 8, seed: int = 0, size: float = 2.5,
                    center=(0.1, -0.05, 0.02)) -> dict:
    """Stand-in for the `local_data` dict datasets/dtu.py:268-291 hands to the model (VisMVSNet feature maps and MVS
    cameras come from an unavailable download): feature maps at half the image resolution for the reference view and
    the other views, `cam` / `src_cams` = [2,4,4] packs (world->camera extrinsic; K in [1,:3,:3]) expressed in the
    de-normalised world frame p_world = p / 2 * size + center that feat_utils.get_local_loss:405-408 maps back to."""
    rng = np.random.default_rng(seed + 4242)
    poses = scene["poses"]
    n_views = len(poses)
    src = [v for v in range(n_views) if v != view]
    h2, w2 = IMG_H // 2, IMG_W // 2
    common = _smooth_field(rng, channels, h2, w2)                            # shared structure, so views correlate
    feats = [(common + 0.5 * _smooth_field(rng, channels, h2, w2)).astype(np.float32) for _ in range(n_views)]
    A = np.eye(4)
    A[:3, :3] *= size / 2.0
    A[:3, 3] = np.asarray(center, np.float64)
    A_inv = np.linalg.inv(A)

    def cam_pack(v):
        pack = np.zeros((2, 4, 4), np.float32)
        pack[0] = (np.linalg.inv(poses[v].astype(np.float64)) @ A_inv).astype(np.float32)
        pack[1] = np.eye(4, dtype=np.float32)
        pack[1, :3, :3] = scene["intrinsics"][:3, :3]
        return pack

    return {"size": np.float32(size), "center": np.asarray(center, np.float32), "feat": feats[view],
            "feat_src": np.stack([feats[v] for v in src]), "cam": cam_pack(view), "src_cams": np.stack([cam_pack(v) for v in src]),
            "H": IMG_H, "W": IMG_W, "src_idxs": np.asarray(src, np.int64)}


def make_raw_cloud(seed: int, n: int = 30000):
    """Unthinned, noisy cloud (what a DUSt3R .ply looks like before load_neural_points thins it, spurfies/model/utils.py:59-88):
    many points per voxel and a few exact duplicates.  -> (pts float32 [n,3], colors uint8 [n,3])."""
    pts, colors, _ = make_cloud(n, spacing=0.008, seed=seed, jitter=0.6)
    rng = np.random.default_rng(seed + 5)
    pts = (pts + 0.004 * rng.standard_normal(pts.shape)).astype(np.float32)
    pts[-50:] = pts[:50]
    return pts, colors.astype(np.uint8)
