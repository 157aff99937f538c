"""Multi-view feature-consistency ("local") loss with the reference's names (spurfies/feat_utils.py:43-77, 377-451).

The reference projects the per-ray SDF zero crossings (PointVolSDF.find_surface_points) into the reference view and its
source views, samples VisMVSNet feature maps there and penalises 1 - cosine similarity.  The feature extractor itself
(feat_utils.py:350-374, needs ckpt/vismvsnet.pt) is outside the hot path: `local_data` arrives with the features already
extracted (datasets/dtu.py:268-291).

The model's training forward runs the term as ONE HIP launch (csrc/local.hip: spf_local_forward — crossing search,
projection, bilinear taps, cosine term, and the tangent the backward needs; ops.local_forward / ops.LocalLoss) fed by a
device-resident descriptor of the view, `local_desc(local_data)`.  The functions with the reference's names below
(`idx_world2cam` ... `get_local_loss`, and the mask-weighted `local_loss_terms`) are the PyTorch formulation with the
reference's signatures: the API mirror for callers that hold arbitrary surface points, and what the GPU tests check the
kernel against at sizes the fixtures do not cover.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import _lib


class LocalDesc:
    """spf_local_desc of one view in device memory (include/spurfies_hip.h): addresses of the [C,H,W] feature maps (reference view
    first, then the sources), the camera packs, size / center and the dimensions.  `buf` uint8 [sizeof]; the kernels read everything
    from it at run time, so a captured hipGraph serves every view by overwriting `buf` (train.py).  `keep`: the tensors the addresses
    point into."""

    NBYTES = ctypes.sizeof(_lib.LocalDesc)

    def __init__(self, buf, n_src, keep=()):
        self.buf, self.n_src, self.keep = buf, int(n_src), tuple(keep)


_DESC_CACHE = OrderedDict()
_DESC_CACHE_MAX = 16


def _ident(v):
    return (v.data_ptr(), v._version, tuple(v.shape), str(v.dtype), str(v.device)) if torch.is_tensor(v) else ("v", repr(v))


def local_desc(local_data, dev) -> LocalDesc:
    """The device descriptor of a `local_data` dict (datasets/dtu.py:268-291: feat [C,H,W], feat_src [m,C,H,W], cam [2,4,4], src_cams
    [m,2,4,4], size, center), built once per distinct set of tensors: the cameras are read back to the host for it (a synchronisation,
    once per view — VolOpt hands the same device tensors for a view every time, train.py:_local_to_device).  Cache entries hold their
    tensors, so an address is never reused while it is a key."""
    if isinstance(local_data, LocalDesc):
        return local_data
    dev = torch.device(dev)
    names = ("feat", "feat_src", "cam", "src_cams", "size", "center")
    key = (str(dev),) + tuple(_ident(local_data[k]) for k in names)
    hit = _DESC_CACHE.get(key)
    if hit is not None:
        _DESC_CACHE.move_to_end(key)
        return hit[0]
    maps = [local_data[k].detach().to(device=dev, dtype=torch.float32).contiguous() for k in ("feat", "feat_src")]
    feat, feat_src = maps
    if feat.dim() != 3 or feat_src.dim() != 4 or feat_src.shape[1:] != feat.shape:
        raise ValueError(f"local_data: feat [C,H,W] / feat_src [m,C,H,W] expected, got {tuple(feat.shape)} / {tuple(feat_src.shape)}")
    m = int(feat_src.shape[0])
    if not 1 <= m < _lib.LOCAL_MAX_VIEWS:
        raise ValueError(f"local_data: {m} source views (the kernel takes 1 .. {_lib.LOCAL_MAX_VIEWS - 1})")
    cams = torch.cat([torch.as_tensor(local_data["cam"]).detach().float().cpu().reshape(1, 2, 4, 4),
                      torch.as_tensor(local_data["src_cams"]).detach().float().cpu().reshape(m, 2, 4, 4)], 0).contiguous()
    d = _lib.LocalDesc()
    plane = feat[0].numel() * feat.shape[0] * 4
    d.feat[0] = feat.data_ptr()
    for s_ in range(m):
        d.feat[1 + s_] = feat_src.data_ptr() + s_ * plane
    flat = cams.reshape(-1).tolist()
    d.cam[: len(flat)] = flat
    d.center[:] = [float(x) for x in torch.as_tensor(local_data["center"]).detach().float().cpu().reshape(3)]
    d.size = float(torch.as_tensor(local_data["size"]).detach().float().cpu().reshape(-1)[0])
    d.n_src, d.C, d.H, d.W = m, int(feat.shape[0]), int(feat.shape[1]), int(feat.shape[2])
    buf = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(dev)
    desc = LocalDesc(buf, m, keep=(feat, feat_src))
    _DESC_CACHE[key] = (desc, [local_data[k] for k in names])
    while len(_DESC_CACHE) > _DESC_CACHE_MAX:
        _DESC_CACHE.popitem(last=False)
    return desc


def idx_world2cam(idx_world_homo, cam):
    """feat_utils.py:43-47: [..., 4, 1] homogeneous world points x cam packs [V,2,4,4] -> camera frame [V, ..., 4, 1]."""
    idx_cam_homo = cam[:, 0:1, ...].unsqueeze(1) @ idx_world_homo
    return idx_cam_homo / (idx_cam_homo[..., -1:, :] + 1e-9)


def idx_cam2img(idx_cam_homo, cam):
    """feat_utils.py:50-55: camera frame -> homogeneous pixel coordinates [V, ..., 3, 1] (K = cam[:, 1, :3, :3])."""
    idx_cam = idx_cam_homo[..., :3, :] / (idx_cam_homo[..., 3:4, :] + 1e-9)
    idx_img_homo = cam[:, 1:2, :3, :3].unsqueeze(1) @ idx_cam
    return idx_img_homo / (idx_img_homo[..., -1:, :] + 1e-9)


def normalize_for_grid_sample(input_, grid):
    """feat_utils.py:58-68: pixel coordinates -> [-1, 1] (clamped to +-1.1) for a [V,C,H,W] map."""
    size = torch.tensor([input_.shape[3], input_.shape[2]], dtype=grid.dtype, device=grid.device).view(1, 1, 1, -1)   # [[[w, h]]]
    return (grid / size * 2 - 1).clamp(-1.1, 1.1)


def get_in_range(grid):
    """feat_utils.py:71-77: 1 where every coordinate of a normalised grid lies in [-1, 1]."""
    return ((grid <= 1) & (grid >= -1)).all(dim=-1).to(grid.dtype)


def _view_terms(pts, weight, feat, cam, feat_src, src_cams, size, center, per_point=False):
    """(sum over sources x points of the masked feature distance, weighted by `weight` [n] in {0,1}; m * sum(weight)).
    per_point: the first value stays per point ([n], summed over the sources only)."""
    pts_world = (pts / 2 * size.reshape(1, 1) + center.reshape(1, 3)).view(1, -1, 1, 3, 1)
    pts_world = torch.cat([pts_world, torch.ones_like(pts_world[..., -1:, :])], dim=-2)
    cam_pack = torch.cat([cam[None], src_cams], dim=0)                                       # [1+m,2,4,4]
    grid = idx_cam2img(idx_world2cam(pts_world, cam_pack), cam_pack)[..., :2, 0]             # [1+m,n,1,2]
    feat_pack = torch.cat([feat[None], feat_src], dim=0)                                     # [1+m,C,H,W]
    grid_n = normalize_for_grid_sample(feat_pack, grid / 2)                                  # the maps are at half resolution
    in_range = get_in_range(grid_n)
    valid = (in_range[:1] * in_range[1:]).unsqueeze(1) > 0.5                                 # [m,1,n,1]
    g = F.grid_sample(feat_pack, grid_n, mode="bilinear", padding_mode="zeros", align_corners=False)
    norm = g.norm(dim=1, keepdim=True)
    corr = (g[:1] * g[1:]).sum(dim=1, keepdim=True) / norm[:1].clamp(min=1e-9) / norm[1:].clamp(min=1e-9)
    corr_loss = (1 - corr).abs()
    keep = valid & (corr_loss < 0.5)
    w = weight.to(corr_loss.dtype).view(1, 1, -1, 1)
    total = torch.where(keep, corr_loss, torch.zeros_like(corr_loss)) * w
    return (total.sum(dim=(0, 1, 3)) if per_point else total.sum()), float(feat_src.shape[0]) * weight.to(corr_loss.dtype).sum()


def local_loss_terms(surf_pts, hit, local_data, per_point=False):
    """surf_pts [R,3] (any finite value where `hit` is False), hit bool [R] -> (sum, count) with local loss = sum / count
    (0 when count == 0), the reference's value for ONE reference view (pointneus_disent.py:744-763).  No host sync."""
    dev = surf_pts.device
    size = torch.as_tensor(local_data["size"], dtype=torch.float32, device=dev)
    center = torch.as_tensor(local_data["center"], dtype=torch.float32, device=dev)
    t = lambda k: local_data[k].to(dev)
    return _view_terms(surf_pts, hit, t("feat"), t("cam"), t("feat_src"), t("src_cams"), size, center, per_point)


def get_local_loss(diff_surf_pts, uncerts, feat, cam, feat_src, src_cams, size, center, network_object_mask, object_mask):
    """feat_utils.py:377-451, reference signature: diff_surf_pts [n_hit,3] = the surface points of the rays where
    `network_object_mask & object_mask` holds, view after view; feat [V,C,H,W], cam [V,2,4,4], feat_src [V,m,C,H,W],
    src_cams [V,m,2,4,4]; the loss is the mean over views of the per-view means."""
    if uncerts is not None:
        raise NotImplementedError("uncertainty-weighted local loss: no shipped config passes uncerts (pointneus_disent.py:753)")
    mask = network_object_mask & object_mask
    if int(mask.sum()) == 0:
        return torch.zeros((), device=diff_surf_pts.device)
    hits = mask.view(feat.shape[0], -1).sum(-1).tolist()
    losses, start = [], 0
    for v, n in enumerate(hits):
        if n > 0:
            pts = diff_surf_pts[start: start + n]
            s, c = _view_terms(pts, torch.ones(n, device=pts.device), feat[v], cam[v], feat_src[v], src_cams[v], size, center)
            losses.append(s / c)
        else:
            losses.append(torch.zeros((), device=diff_surf_pts.device))
        start += n
    return sum(losses) / len(losses)
