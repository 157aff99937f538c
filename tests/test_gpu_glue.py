"""GPU: the small fused kernels around the MLPs — spf_camera_rays against rend_util.get_camera_params (the reference's
formulation, utils/rend_util.py:60-95,143-156) and spf_loss_forward/backward against VolSDFLoss + the pseudo-point term in
PyTorch ops (model/loss.py:42-101, pointneus_disent.py:765-780), including the clipped-mask and no-valid-point branches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_camera_rays_match_the_pytorch_formulation():
    from spurfies_amd import ops, synthetic as syn
    from spurfies_amd.utils import rend_util

    K4, poses = syn.make_cameras()
    g = torch.Generator().manual_seed(0)
    uv = torch.from_numpy(syn.make_pixels(777, g))[None].cuda()
    for K in (torch.from_numpy(K4)[None].cuda(), torch.from_numpy(K4[:3, :3].copy())[None].cuda()):
        for p in range(3):
            pose = torch.from_numpy(poses[p])[None].cuda()
            dirs, loc, scale = ops.camera_rays(uv, pose, K)
            d_ref, loc_ref = rend_util.get_camera_params(uv, pose, K)
            d_cam, _ = rend_util.get_camera_params(uv, torch.eye(4, device="cuda")[None], K)
            np.testing.assert_allclose(dirs.cpu().numpy(), d_ref[0].cpu().numpy(), rtol=0, atol=3e-7)
            np.testing.assert_allclose(loc.cpu().numpy(), loc_ref.expand(777, 3).cpu().numpy(), rtol=0, atol=0)
            np.testing.assert_allclose(scale.cpu().numpy(), d_cam[0, :, 2:].cpu().numpy(), rtol=0, atol=3e-7)
    assert ops.camera_rays(uv.repeat(2, 1, 1), pose.repeat(2, 1, 1), K.repeat(2, 1, 1)) is None      # multi-view batches: caller's PyTorch path


def _torch_loss(rgb, acc, psdf, tv, grad, slot_valid, pvalid, ray_valid, rgb_gt, mask, w):
    import torch.nn.functional as F

    out = {"rgb": (rgb - rgb_gt).abs().mean()}
    out["mask"] = F.binary_cross_entropy(acc.clip(1e-3, 1.0 - 1e-3), mask)
    g = grad[slot_valid.bool()]
    out["eik"] = ((g.norm(2, dim=1) - 1) ** 2).mean() if g.shape[0] else torch.zeros((), device=rgb.device)
    use = pvalid.bool() & ray_valid.bool()
    out["pseudo"] = psdf[use].abs().mean() if bool(use.any()) else torch.full((), 1000.0, device=rgb.device)
    out["loss"] = w["rgb"] * out["rgb"] + w["eik"] * out["eik"] + w["tv"] * tv + w["pseudo"] * out["pseudo"] + out["mask"]
    return out


@pytest.mark.parametrize("any_pseudo", [True, False])
def test_fused_loss_matches_pytorch_ops(any_pseudo):
    from spurfies_amd import _lib, ops

    g = torch.Generator().manual_seed(3)
    R, rows = 333, 333 * 80
    rgb = torch.rand((R, 3), generator=g).cuda().requires_grad_(True)
    rgb_gt = torch.rand((R, 3), generator=g).cuda()
    acc_v = torch.rand((R,), generator=g) * 1.2 - 0.1                       # some values outside the clip interval [1e-3, 1 - 1e-3]
    acc = acc_v.cuda().requires_grad_(True)
    mask3 = (torch.rand((R,), generator=g) > 0.3).float()[:, None].repeat(1, 3).cuda()
    psdf = (torch.randn((R,), generator=g) * 0.1).cuda().requires_grad_(True)
    tv = torch.tensor(0.37, device="cuda", requires_grad=True)
    grad = torch.randn((rows, 3), generator=g).cuda()
    slot_valid = (torch.rand((rows,), generator=g) > 0.4).to(torch.uint8).cuda()
    n_points = slot_valid.sum().to(torch.int32).reshape(1)
    pvalid = (torch.rand((R,), generator=g) > 0.5).to(torch.uint8).cuda() if any_pseudo else torch.zeros((R,), dtype=torch.uint8, device="cuda")
    ray_valid = (torch.rand((R,), generator=g) > 0.2).to(torch.uint8).cuda()
    w = dict(rgb=1.0, eik=0.001, tv=0.01, pseudo=0.5)
    lw = _lib.LossWeights(w["rgb"], w["eik"], w["tv"], 0.5, w["pseudo"], 1)
    total, terms = ops.FusedLoss.apply(rgb, acc, psdf, tv, grad, slot_valid, n_points, pvalid, ray_valid, rgb_gt, mask3, mask3.stride(0), lw, None)
    total.backward()
    got = {k: t.grad.clone() for k, t in (("rgb", rgb), ("acc", acc), ("psdf", psdf), ("tv", tv))}
    for t in (rgb, acc, psdf, tv):
        t.grad = None
    ref = _torch_loss(rgb, acc, psdf, tv, grad, slot_valid, pvalid, ray_valid, rgb_gt, mask3[:, 0], w)
    ref["loss"].backward()
    np.testing.assert_allclose(total.item(), ref["loss"].item(), rtol=2e-6)
    for i, k in ((1, "rgb"), (2, "eik"), (4, "mask"), (6, "pseudo")):
        np.testing.assert_allclose(terms[i].item(), ref[k].item(), rtol=3e-6, err_msg=k)
    np.testing.assert_allclose(terms[3].item(), 0.37, rtol=1e-6)
    for k, t in (("rgb", rgb), ("acc", acc), ("tv", tv)):
        np.testing.assert_allclose(got[k].cpu().numpy(), t.grad.cpu().numpy(), rtol=2e-5, atol=1e-9, err_msg=k)
    ref_p = psdf.grad if psdf.grad is not None else torch.zeros_like(psdf)
    np.testing.assert_allclose(got["psdf"].cpu().numpy(), ref_p.cpu().numpy(), rtol=2e-5, atol=1e-9)
    assert float(got["acc"][(acc_v < 1e-3).cuda() | (acc_v > 1 - 1e-3).cuda()].abs().max()) == 0.0     # clamp passes no gradient outside
