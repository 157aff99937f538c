"""ORACLE (test infrastructure, never shipped): CPU restatement, in plain PyTorch-CPU
ops, of the reference's per-ray volumetric-rendering hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (spurfies_amd/) never imports anything under oracle/.

Pinning: every function here is checked against outputs of the reference's own Python
(imported from /root/reference in the authoring container by oracle/make_golden.py) through
the fixtures in tests/golden/*.npz — see tests/test_oracle_golden.py.  The one stage that
cannot be pinned is the kNN op itself (upstream torch_knnquery source absent): see
oracle/voxel_grid.py.

Each function cites the reference lines it follows (paths relative to /root/reference).
The arithmetic is kept op-for-op (same torch ops, same association order) so that CPU
results agree with the reference to float32 round-off.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from oracle.voxel_grid import VoxelGridOracle

SDF_FILL = 1000.0  # pointneus_disent.py:271,371,445,703


# --------------------------------------------------------------------------- config
@dataclass
class PathConfig:
    """Effective values of config/vol/dtu_pn.yaml:23-44 + config/ours.yaml:9-24."""
    k: int = 8
    r: float = 2.0
    rbf: float = 45.0                       # pointneus_disent.py:42
    max_shading_pts: int = 80
    scene_bounding_sphere: float = 3.0
    beta_min: float = 1e-4
    near: float = 0.5
    far_cfg: float = 4.5                    # only the fill value of depth_vals (pointneus_disent.py:833-840)
    n_samples: int = 64
    n_samples_eval: int = 128
    n_samples_extra: int = 32
    eps: float = 0.1
    beta_iters: int = 10
    max_total_iters: int = 5
    add_tiny: float = 0.0
    voxel_size: tuple = (0.025, 0.025, 0.025)   # pointneus_disent.py:45-62
    voxel_scale: tuple = (3, 3, 3)
    kernel_size: tuple = (3, 3, 3)
    max_points_per_voxel: int = 26
    max_occ_voxels: int = 20000
    ranges: tuple = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
    # loss weights, config/ours.yaml:15-20
    rgb_weight: float = 1.0
    eikonal_weight: float = 0.001
    tv_weight: float = 0.01
    local_weight: float = 0.5
    pseudo_weight: float = 0.5

    @property
    def far(self):  # ray_sampler.py:353 -> RaySampler(near, 2*scene_bounding_sphere); conf far is ignored
        return 2.0 * self.scene_bounding_sphere


MLP_LAYERS = {"F_color": (0, 2, 4, 6), "F_geometry": (0, 2, 4, 6, 8), "T": (0,), "R": (0, 2, 4)}
TRAINABLE = ("neural_feats_color", "neural_feats_geometry", "F_color", "R", "density.beta")


def load_state(state: dict, requires_grad: bool = True) -> dict:
    """numpy/torch state_dict -> dict of float32 CPU tensors; trainable ones are leaves with
    grad as train.py:145-154 leaves them (F_geometry / T frozen)."""
    out = {}
    for k, v in state.items():
        t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).detach().clone().float()
        trainable = requires_grad and any(k == n or k.startswith(n + ".") for n in TRAINABLE)
        out[k] = t.requires_grad_(trainable)
    return out


def make_grid(cfg: PathConfig, neural_pts) -> VoxelGridOracle:
    g = VoxelGridOracle(cfg.voxel_size, cfg.voxel_scale, cfg.kernel_size, cfg.max_points_per_voxel,
                        cfg.max_occ_voxels, cfg.ranges)
    g.set_pointset(neural_pts.detach().numpy())
    return g


# --------------------------------------------------------------------------- small pieces
def mlp(x, st, name):
    """nn.Sequential(Linear, LeakyReLU, ..., Linear) of pointneus_disent.py:76-107 (slope 0.01)."""
    pos = MLP_LAYERS[name]
    for i, p in enumerate(pos):
        x = F.linear(x, st[f"{name}.{p}.weight"], st[f"{name}.{p}.bias"])
        if i + 1 < len(pos):
            x = F.leaky_relu(x, 0.01)
    return x


def posenc(x, n_freqs):
    """embedder.py:5-49: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]."""
    outs = [x]
    for f in 2.0 ** torch.linspace(0.0, n_freqs - 1, n_freqs):
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def get_beta(st, cfg):
    """density.py:28-30."""
    return st["density.beta"].abs() + torch.tensor(cfg.beta_min)


def laplace_density(sdf, beta):
    """density.py:21-26."""
    alpha = 1 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def camera_rays(uv, pose, intrinsics):
    """rend_util.py:60-95 + lift :143-156.  uv [1,R,2], pose [1,4,4], K [1,4,4] -> dirs [1,R,3], cam_loc [1,3]."""
    cam_loc = pose[:, :3, 3]
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    cam_pts = torch.stack((x_lift, y_lift, z), dim=-1).permute(0, 2, 1)
    world = (torch.bmm(pose[:, :3, :3], cam_pts) + pose[:, :3, 3:]).permute(0, 2, 1)
    dirs = F.normalize(world - cam_loc[:, None, :], dim=2)
    return dirs, cam_loc


def volume_weights(deltas, density):
    """pointneus_disent.py:894-908."""
    free_energy = deltas * density
    shifted = torch.cat([torch.zeros(deltas.shape[0], 1), free_energy[:, :-1]], dim=-1)
    alpha = 1 - torch.exp(-free_energy)
    transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
    return alpha * transmittance


# --------------------------------------------------------------------------- kNN glue
def knn_query(grid: VoxelGridOracle, pts, k, r, sr):
    """spurfies/model/utils.py:90-113 on top of the op.  pts [R,D,3] ->
    neighbor_idx int64 [P,k] (-1 pad), shading_pts [P,3], mask bool [R,SR], ray_mask bool [R]."""
    R = pts.shape[0]
    pidx, loc, _, ray_valid = grid.query_dense(pts.detach().numpy(), k, r, sr)
    pidx_t = torch.from_numpy(pidx).long()
    pt_valid = (pidx_t >= 0).any(-1)                       # [R,SR]
    mask = pt_valid & torch.from_numpy(ray_valid)[:, None]  # rows of invalid rays are all False anyway
    neighbor_idx = pidx_t[mask]
    shading_pts = torch.from_numpy(loc)[mask]
    return neighbor_idx, shading_pts, mask.view(R, -1), torch.from_numpy(ray_valid)


def pair_index(valid):
    """utils.py:172-183: row id of every valid (point, neighbour) pair, grouped by point."""
    return torch.arange(valid.shape[0]).view(-1, 1).expand_as(valid)[valid]


def gather_pairs(neighbor_idx, valid, st):
    """utils.py:140-170 (concat -> index_select -> masked_select), returned already split."""
    idx = neighbor_idx.clone()
    idx[~valid] = 0
    flat = idx[valid]
    return st["neural_pts"][flat], st["neural_feats_color"][flat], st["neural_feats_geometry"][flat]


def rbf_weights(x_pi, pair_row, n_pts, rbf):
    """pointneus_disent.py:241-247 (weights are detached)."""
    dist = torch.clamp(torch.norm(x_pi, dim=-1), min=1e-12).clone().detach()
    w = torch.exp(-((dist * rbf) ** 2))
    norm = torch.zeros(n_pts).index_add_(0, pair_row, w)
    return w, norm


def sdf_pairs_to_points(x_pi, feat_geo, w, norm, pair_row, n_pts, st):
    """pointneus_disent.py:300-313: input order [geometry latent (32) | x_pi (3)]."""
    sdf = mlp(mlp(torch.cat([feat_geo, x_pi], dim=-1), st, "F_geometry"), st, "T")
    acc = torch.zeros(n_pts, 1).index_add_(0, pair_row, w.unsqueeze(-1) * sdf)
    return acc / norm.unsqueeze(-1)


def sdf_at_points(x, grid, st, cfg):
    """pointneus_disent.py:249-298 (get_sdf_eval) == :348-421 (sdf_importance) == :423-495
    (pseudo_sdf).  x [M,3] -> (sdf [M] with 1000 where no neighbour, valid bool [M]).
    Differentiable w.r.t. x and the latents when called outside no_grad."""
    nb, _, mask, ray_mask = knn_query(grid, x.unsqueeze(1), cfg.k, cfg.r, 1)
    filler = torch.ones(x.shape[0]) * SDF_FILL
    if nb.shape[0] == 0:
        return filler, ray_mask
    valid = nb >= 0
    rows = pair_index(valid)
    pos, _, fg = gather_pairs(nb, valid, st)
    pts = x.unsqueeze(1)[mask]
    x_pi = pts[rows, :] - pos
    w, norm = rbf_weights(x_pi, rows, nb.shape[0], cfg.rbf)
    agg = sdf_pairs_to_points(x_pi, fg, w, norm, rows, nb.shape[0], st)
    out = filler.clone()
    out[ray_mask] = agg.squeeze(-1)
    return out, ray_mask


# --------------------------------------------------------------------------- sampler
def uniform_z(n_rays, cfg, training, draws=None):
    """ray_sampler.py:33-59 (UniformSampler, N = N_samples_eval, far = 2*sphere)."""
    near = cfg.near * torch.ones(n_rays, 1)
    far = cfg.far * torch.ones(n_rays, 1)
    t = torch.linspace(0.0, 1.0, steps=cfg.n_samples_eval)
    z = near * (1.0 - t) + far * t
    if training:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        t_rand = _draw(draws, "uniform_rand", lambda: torch.rand(z.shape))
        z = lower + (upper - lower) * t_rand
    return z


def _draw(draws, key, fn):
    """RNG draws come from the CPU generator in the reference (ray_sampler.py:55,514,550,562).
    `draws` (dict) records them when empty and replays them when filled."""
    if draws is None:
        return fn()
    if key in draws:
        return torch.as_tensor(draws[key])
    v = fn()
    draws[key] = v.clone()
    return v


def error_bound(beta, sdf, z, dists, d_star):
    """ray_sampler.py:576-588."""
    density = laplace_density(sdf.reshape(z.shape), beta)
    shifted = torch.cat([torch.zeros(dists.shape[0], 1), dists * density[:, :-1]], dim=-1)
    integral = torch.cumsum(shifted, dim=-1)
    err = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
    err_int = torch.cumsum(err, dim=-1)
    bound = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bound.max(-1)[0]


def d_star_of(z, d):
    """ray_sampler.py:417-432 (Theorem 1 bound per interval)."""
    dists = z[:, 1:] - z[:, :-1]
    a, b, c = dists, d[:, :-1].abs(), d[:, 1:].abs()
    first = a.pow(2) + b.pow(2) <= c.pow(2)
    second = a.pow(2) + c.pow(2) <= b.pow(2)
    d_star = torch.zeros(z.shape[0], z.shape[1] - 1)
    d_star[first] = b[first]
    d_star[second] = c[second]
    s = (a + b + c) / 2.0
    area = s * (s - a) * (s - b) * (s - c)
    m = ~first & ~second & (b + c - a > 0)
    d_star[m] = (2.0 * torch.sqrt(area[m])) / (a[m])
    d_star = (d[:, 1:].sign() * d[:, :-1].sign() == 1) * d_star
    return dists, d_star


def invert_cdf(cdf, bins, u):
    """ray_sampler.py:517-529."""
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def error_bounded_z(ray_dirs, cam_loc, grid, st, cfg, training, fast=-1, draws=None, trace=None):
    """ray_sampler.py:377-574 (ErrorBoundSampler_pn.get_z_vals).  ray_dirs, cam_loc [R,3] -> z [R,98].
    `trace` (dict) receives per-iteration intermediates for stage tests."""
    max_iters = fast if fast >= 0 else cfg.max_total_iters
    beta0 = get_beta(st, cfg).detach()
    R = ray_dirs.shape[0]
    z = uniform_z(R, cfg, training, draws)
    samples, samples_idx = z, None
    dists = z[:, 1:] - z[:, :-1]
    bound = (1.0 / (4.0 * torch.log(torch.tensor(cfg.eps + 1.0)))) * (dists ** 2.0).sum(-1)
    beta = torch.sqrt(bound)
    iters, not_converge = 0, True
    sdf = None
    while not_converge and iters < max_iters:
        pts = cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)
        with torch.no_grad():
            s_sdf, _ = sdf_at_points(pts.reshape(-1, 3), grid, st, cfg)
        if samples_idx is not None:
            merged = torch.cat([sdf.reshape(-1, z.shape[1] - samples.shape[1]),
                                s_sdf.reshape(-1, samples.shape[1])], -1)
            sdf = torch.gather(merged, 1, samples_idx).reshape(-1, 1)
        else:
            sdf = s_sdf
        d = sdf.reshape(z.shape)
        dists, d_star = d_star_of(z, d)
        err = error_bound(beta0, sdf, z, dists, d_star)
        beta[err <= cfg.eps] = beta0
        b_min, b_max = beta0.unsqueeze(0).repeat(R), beta
        for _ in range(cfg.beta_iters):
            b_mid = (b_min + b_max) / 2.0
            err = error_bound(b_mid.unsqueeze(-1), sdf, z, dists, d_star)
            b_max[err <= cfg.eps] = b_mid[err <= cfg.eps]
            b_min[err > cfg.eps] = b_mid[err > cfg.eps]
        beta = b_max
        density = laplace_density(sdf.reshape(z.shape), beta.unsqueeze(-1))
        dists_inf = torch.cat([dists, torch.tensor([1e10]).unsqueeze(0).repeat(R, 1)], -1)
        free = dists_inf * density
        shifted = torch.cat([torch.zeros(R, 1), free[:, :-1]], dim=-1)
        alpha = 1 - torch.exp(-free)
        trans = torch.exp(-torch.cumsum(shifted, dim=-1))
        weights = alpha * trans
        iters += 1
        not_converge = bool(beta.max() > beta0)
        more = not_converge and iters < max_iters
        if more:
            N = cfg.n_samples_eval
            err_sec = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists_inf[:, :-1] ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
            err_int = torch.cumsum(err_sec, dim=-1)
            bound_op = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * trans[:, :-1]
            pdf = bound_op + cfg.add_tiny
        else:
            N = cfg.n_samples
            pdf = weights[..., :-1] + 1e-5
        pdf = pdf / torch.sum(pdf, -1, keepdim=True)
        cdf = torch.cumsum(pdf, -1)
        cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
        if more or not training:
            u = torch.linspace(0.0, 1.0, steps=N).unsqueeze(0).repeat(R, 1)
        else:
            u = _draw(draws, "cdf_rand", lambda: torch.rand([R, N]))
        u = u.contiguous()
        if trace is not None:
            trace[f"it{iters}"] = dict(z=z.clone(), sdf=d.clone(), d_star=d_star.clone(), beta=beta.clone(),
                                       weights=weights.clone(), cdf=cdf.clone(), u=u.clone())
        samples = invert_cdf(cdf, z, u)
        if more:
            z, samples_idx = torch.sort(torch.cat([z, samples], -1), -1)
    z_samples = samples
    near = cfg.near * torch.ones(R, 1)
    far = cfg.far * torch.ones(R, 1)
    if cfg.n_samples_extra > 0:
        if training:
            sel = _draw(draws, "extra_perm", lambda: torch.randperm(z.shape[1]))[: cfg.n_samples_extra]
        else:
            sel = torch.linspace(0, z.shape[1] - 1, cfg.n_samples_extra).long()
        extra = torch.cat([near, far, z[:, sel]], -1)
    else:
        extra = torch.cat([near, far], -1)
    z_out, _ = torch.sort(torch.cat([z_samples, extra], -1), -1)
    _draw(draws, "eik_idx", lambda: torch.randint(z_out.shape[-1], (R,)))  # ray_sampler.py:562 (value unused)
    if trace is not None:
        trace["iters"] = iters
    return z_out


# --------------------------------------------------------------------------- tv
def tv_loss(grid, st, cfg):
    """spurfies/model/utils.py:221-282."""
    kp_pos = st["neural_pts"].detach()
    kp_feat = st["neural_feats_geometry"]
    n = kp_pos.shape[0]
    nb, _, kmask, _ = knn_query(grid, kp_pos.view(n, 1, 3), cfg.k, cfg.r, 1)
    padded = torch.full((n, cfg.k), -1, dtype=torch.long)
    padded[..., 0] = torch.arange(n)
    padded.masked_scatter_(kmask, nb)
    origin = torch.arange(n)[:, None]
    ident = padded == origin
    enough = (padded >= 0).int().sum(dim=-1, keepdim=True) > 1
    padded[ident & enough] = -1
    valid = padded >= 0
    rows = pair_index(valid)
    flat = padded[valid]
    w = 1 / (torch.linalg.norm(kp_pos[flat] - kp_pos[rows], dim=-1) + 1.0e-5)
    norm = torch.zeros(n).index_add_(0, rows, w)
    fdist = torch.linalg.norm(kp_feat[flat] - kp_feat[rows], ord=1, dim=-1)
    tv = torch.zeros(n).index_add_(0, rows, w * fdist)
    return (tv / norm).mean()


# --------------------------------------------------------------------------- local (feature-consistency) loss
def find_surface_points(sdf, z):
    """pointneus_disent.py:586-612: per ray, the FIRST interval whose SDF changes sign from + to - (back-facing crossings are
    ignored; slots without a point carry 1000 -> NaN and never form a crossing), linearly interpolated depth.
    sdf, z [Rv,SR] -> (d_surface [Rv] (0 where no crossing), hit bool [Rv])."""
    s = torch.where(sdf == SDF_FILL, torch.full_like(sdf, float("nan")), sdf)
    cross = (s[:, 1:] * s[:, :-1] < 0) & (s[:, 1:] < s[:, :-1])
    hit = cross.any(-1)
    first = ((torch.cumsum(cross.int(), -1) == 0).sum(-1)).clamp(max=cross.shape[1] - 1)      # index of the first True
    s0, s1 = torch.gather(s, 1, first[:, None]).squeeze(1), torch.gather(s, 1, first[:, None] + 1).squeeze(1)
    d0, d1 = torch.gather(z, 1, first[:, None]).squeeze(1), torch.gather(z, 1, first[:, None] + 1).squeeze(1)
    d = torch.zeros(sdf.shape[0])
    d[hit] = ((s0 * d1 - s1 * d0) / (s0 - s1))[hit]
    return d, hit


def local_feature_loss(surf_pts, local):
    """feat_utils.py:377-451 for one reference view, uncerts=None: project the surface points into the reference and the
    source views (idx_world2cam :43-47, idx_cam2img :50-55), bilinear-sample the half-resolution feature maps
    (normalize_for_grid_sample :58-68, grid / 2), 1 - cosine similarity between the reference and each source feature,
    masked to points visible in both views and to |1 - corr| < 0.5, mean over sources x points.
    surf_pts [n,3] (n > 0); local: size [], center [3], feat [C,H,W], feat_src [m,C,H,W], cam [2,4,4], src_cams [m,2,4,4]."""
    n = surf_pts.shape[0]
    if n == 0:
        return torch.tensor(0.0)
    pw = (surf_pts / 2 * local["size"].view(1, 1) + local["center"].view(1, 3)).view(1, -1, 1, 3, 1)
    pw = torch.cat([pw, torch.ones_like(pw[..., -1:, :])], dim=-2)                          # [1,n,1,4,1]
    cams = torch.cat([local["cam"][None], local["src_cams"]], dim=0)                        # [1+m,2,4,4]
    pc = cams[:, 0:1].unsqueeze(1) @ pw
    pc = pc / (pc[..., -1:, :] + 1e-9)
    pc3 = pc[..., :3, :] / (pc[..., 3:4, :] + 1e-9)
    pi = cams[:, 1:2, :3, :3].unsqueeze(1) @ pc3
    pi = pi / (pi[..., -1:, :] + 1e-9)
    grid = pi[..., :2, 0]                                                                   # [1+m,n,1,2] pixel (x, y)
    feats = torch.cat([local["feat"][None], local["feat_src"]], dim=0)                      # [1+m,C,H,W]
    wh = torch.tensor(feats.shape[2:]).flip(0).to(grid.dtype).view(1, 1, 1, -1)
    gn = ((grid / 2) / wh * 2 - 1).clamp(-1.1, 1.1)
    in_range = ((gn[..., 0] <= 1) & (gn[..., 0] >= -1) & (gn[..., 1] <= 1) & (gn[..., 1] >= -1)).to(gn.dtype)
    valid = (in_range[:1] * in_range[1:]).unsqueeze(1) > 0.5                                # [m,1,n,1]
    g = F.grid_sample(feats, gn, mode="bilinear", padding_mode="zeros", align_corners=False)  # [1+m,C,n,1]
    gnorm = g.norm(dim=1, keepdim=True)
    corr = (g[:1] * g[1:]).sum(dim=1, keepdim=True) / gnorm[:1].clamp(min=1e-9) / gnorm[1:].clamp(min=1e-9)
    corr_loss = (1 - corr).abs()
    return (corr_loss * valid * (corr_loss < 0.5)).mean()


# --------------------------------------------------------------------------- full forward
def forward(inp, st, cfg: PathConfig, grid=None, training=True, fast=-1, draws=None, stages=None):
    """pointneus_disent.py:614-892.  inp: {'intrinsics' [1,4,4], 'uv' [1,R,2], 'pose' [1,4,4]}.
    Returns the reference's output dict; `stages` (dict) receives intermediates."""
    if grid is None:
        grid = make_grid(cfg, st["neural_pts"])
    K, uv, pose = inp["intrinsics"], inp["uv"], inp["pose"]
    pseudo_loss = torch.tensor(0.0)
    local_loss = torch.tensor(0.0)
    local = inp.get("local_data")
    dirs_b, cam_b = camera_rays(uv, pose, K)
    dirs_tmp, _ = camera_rays(uv, torch.eye(4)[None], K)
    depth_scale = dirs_tmp[0, :, 2:]
    ray_dirs = dirs_b.reshape(-1, 3)
    R = ray_dirs.shape[0]
    cam_loc = cam_b.unsqueeze(1).repeat(1, R, 1).reshape(-1, 3)
    trace = {} if stages is not None else None
    z = error_bounded_z(ray_dirs, cam_loc, grid, st, cfg, training, fast, draws, trace)
    points = cam_loc.unsqueeze(1) + z.unsqueeze(2) * ray_dirs.unsqueeze(1)
    nb, sh_pts, mask, ray_mask = knn_query(grid, points, cfg.k, cfg.r, cfg.max_shading_pts)
    vmask = mask[ray_mask]                                   # [Rv,SR]
    SR = cfg.max_shading_pts
    have = sh_pts.shape[0] > 0
    grads = None
    if have:
        # filter_points, pointneus_disent.py:207-239
        o, dvec = cam_loc[ray_mask], ray_dirs[ray_mask]
        sqp = torch.zeros((*vmask.shape, 3))
        sqp[vmask] = sh_pts.clone().detach()
        t = ((sqp - o.unsqueeze(1)) / dvec.unsqueeze(1)).nanmean(dim=-1, keepdim=True)
        zf = torch.zeros_like(t)
        zf[vmask] = t[vmask]
        zpad = torch.cat([zf, torch.zeros(zf.shape[0], 1, 1)], dim=1)
        deltas = zpad[:, 1:] - zpad[:, :-1]
        deltas[~vmask] = 0
        deltas = deltas.clamp_(min=0)
        pts = (o.unsqueeze(1) + zf * dvec.unsqueeze(1))[vmask]
        pts = pts.detach().requires_grad_(True)
        valid = nb >= 0
        n_pts = nb.shape[0]
        rows = pair_index(valid)
        pos, fc, fg = gather_pairs(nb, valid, st)
        x_pi = pts[rows, :] - pos
        w, norm = rbf_weights(x_pi, rows, n_pts, cfg.rbf)
        agg_sdf = sdf_pairs_to_points(x_pi, fg, w, norm, rows, n_pts, st)
        grads = torch.autograd.grad(agg_sdf, pts, torch.ones_like(agg_sdf), retain_graph=True, create_graph=True)[0]
        # get_color, pointneus_disent.py:325-346
        feat = mlp(torch.cat([posenc(x_pi, 6), fc], dim=-1), st, "F_color")
        acc = torch.zeros(n_pts, 256).index_add_(0, rows, w.unsqueeze(-1) * feat)
        agg_feat = acc / norm.unsqueeze(-1)
        dir_pts = dvec.unsqueeze(1).expand(-1, SR, -1)[vmask]
        colors = torch.sigmoid(mlp(torch.cat([posenc(dir_pts, 3), agg_feat], dim=-1), st, "R"))
        sdf_f = torch.ones_like(deltas) * SDF_FILL
        sdf_f[vmask] = agg_sdf
        dens = torch.zeros_like(deltas)
        dens[vmask] = laplace_density(agg_sdf, get_beta(st, cfg))
        weights_v = volume_weights(deltas[..., 0], dens[..., 0])
        zv = zf.squeeze(-1)
        if local is not None and training:              # pointneus_disent.py:727-763
            d_surf, hit = find_surface_points(sdf_f.squeeze(-1), zv)
            local_loss = local_feature_loss((o + dvec * d_surf[:, None])[hit], local)
            if stages is not None:
                stages.update(d_surface=d_surf, network_mask=hit)
        dist_map = torch.sum(weights_v / (weights_v.sum(-1, keepdim=True) + 1e-10) * zv, -1)
        pts_rendered = o + dvec * dist_map[:, None]
        sdf_r, rend_valid = sdf_at_points(pts_rendered, grid, st, cfg)
        pseudo_vals = _pseudo_values(sdf_r, rend_valid)
        pseudo_loss = F.l1_loss(pseudo_vals, torch.zeros_like(pseudo_vals), reduction="mean")
        col_f = torch.zeros((*deltas.shape[:2], 3))
        col_f[vmask] = colors
        rgb_v = torch.sum(weights_v.unsqueeze(-1) * col_f, 1)
        depth_v = torch.sum(weights_v * zv, 1, keepdim=True) / (weights_v.sum(dim=1, keepdim=True) + 1e-8)
        acc_v = torch.sum(weights_v, -1, keepdim=True)
        if not training:
            nrm = torch.zeros((*deltas.shape[:2], 3))
            g = grads.detach()
            nrm[vmask] = g / g.norm(2, -1, keepdim=True)
            normal_v = torch.sum(weights_v.unsqueeze(-1) * nrm, 1)
        pts_f = torch.zeros((*deltas.shape[:2], 3))
        pts_f[vmask] = pts
        if stages is not None:
            stages.update(z=z, points=points, neighbor_idx=nb, mask=mask, ray_mask=ray_mask, shading_pts=pts,
                          z_slots=zf, deltas=deltas, x_pi=x_pi, w=w, norm=norm, agg_sdf=agg_sdf, grads=grads,
                          colors=colors, weights=weights_v, dist_map=dist_map, sdf_rendered=sdf_r, trace=trace)
    rgb = torch.zeros((R, 3))
    normal = torch.zeros((R, 3))
    depth = torch.full((R, 1), 1, dtype=torch.float32)
    weights = torch.zeros((R, SR))
    depth_vals = torch.ones((R, SR)) * cfg.far_cfg
    xyz = torch.zeros((R, SR, 3))
    if have:
        xyz[ray_mask] = pts_f
        rgb[ray_mask] = rgb_v
        depth[ray_mask] = depth_v
        weights[ray_mask] = weights_v
        depth_vals[ray_mask] = zv * depth_scale[ray_mask]
    out = {"rgb_values": rgb, "depth_values": depth, "depth_vals": depth_vals, "weights": weights, "xyz": xyz,
           "local_loss": local_loss, "pseudo_pts_loss": pseudo_loss, "tv_loss": tv_loss(grid, st, cfg)}
    if not training:
        if have:
            normal[ray_mask] = normal_v
        out["normal_map"] = normal
    else:
        out["grad_theta"] = grads
    return out


def _pseudo_values(sdf_filled, valid):
    """pointneus_disent.py:445-495: pseudo_sdf returns only the valid rows' SDF, or a constant
    1000-vector over ALL inputs when no input has a neighbour."""
    if not bool(valid.any()):
        return torch.ones(sdf_filled.shape[0]) * SDF_FILL
    return sdf_filled[valid]


# --------------------------------------------------------------------------- loss
def volsdf_loss(out, rgb_gt, mask_gt, cfg: PathConfig):
    """spurfies/model/loss.py:51-101 with the weights of config/ours.yaml:15-20.
    rgb_gt [R,3]; mask_gt [R] in {0,1} (loss.py:71 takes mask[:, 0])."""
    res = {"rgb_loss": F.l1_loss(out["rgb_values"], rgb_gt.reshape(-1, 3), reduction="mean")}
    g = out.get("grad_theta")
    res["eikonal_loss"] = ((g.norm(2, dim=1) - 1) ** 2).mean() if g is not None else torch.tensor(0.0)
    res["tv_loss"] = out["tv_loss"] if cfg.tv_weight > 0 else torch.tensor(0.0)
    wsum = out["weights"].sum(-1, keepdim=True)
    res["mask_loss"] = F.binary_cross_entropy(wsum.clip(1e-3, 1.0 - 1e-3), mask_gt.reshape(-1, 1).float())
    res["local_loss"] = out["local_loss"]
    res["pseudo_loss"] = out["pseudo_pts_loss"] if cfg.pseudo_weight > 0 else torch.tensor(0.0)
    res["loss"] = (cfg.rgb_weight * res["rgb_loss"] + cfg.eikonal_weight * res["eikonal_loss"]
                   + cfg.tv_weight * res["tv_loss"] + cfg.local_weight * res["local_loss"]
                   + cfg.pseudo_weight * res["pseudo_loss"] + res["mask_loss"])
    return res


def make_optimizer(st, lr=5.0e-4):
    """spurfies/train.py:168-189: Adam, two param groups (the first one empty), CosineAnnealingLR(T_max=1e5, eta_min=3e-4)."""
    params = [v for v in st.values() if v.requires_grad]
    opt = torch.optim.Adam([{"params": [], "lr": 1e-2}, {"params": params, "lr": lr}])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100_000, eta_min=3e-4, last_epoch=-1)
    return opt, sched


def optimizer_step(st, opt, sched=None, max_norm=1.0):
    """spurfies/train.py:359-363, 548-564: clip_grad_norm_(1.0), drop the update when a gradient is not finite, Adam, scheduler.
    Returns the total gradient norm before clipping."""
    params = [v for v in st.values() if v.requires_grad]
    norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
    if all(bool(torch.isfinite(p.grad).all()) for p in params if p.grad is not None):
        opt.step()
    if sched is not None:
        sched.step()
    return norm


def train_step_grads(inp, rgb_gt, mask_gt, st, cfg, grid=None, draws=None, stages=None):
    """One forward + loss + backward (spurfies/train.py:330-361, without the optimiser).
    Returns (outputs, losses, {name: grad}) for the trainable tensors."""
    for v in st.values():
        if v.requires_grad and v.grad is not None:
            v.grad = None
    out = forward(inp, st, cfg, grid=grid, training=True, fast=1, draws=draws, stages=stages)
    losses = volsdf_loss(out, rgb_gt, mask_gt, cfg)
    losses["loss"].backward()
    grads = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in st.items() if v.requires_grad}
    return out, losses, grads
