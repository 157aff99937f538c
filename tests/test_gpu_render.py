"""GPU parity: per-ray compositing kernels vs the oracle's torch-CPU restatement
(filter_points pointneus_disent.py:207-239, LaplaceDensity density.py:16-30, volume_rendering :894-908,
composites :765-795) — forward values and every gradient."""
import numpy as np
import pytest
import torch

from oracle import path as P

pytestmark = pytest.mark.gpu


def _inputs(R=300, SR=80, seed=0):
    g = torch.Generator().manual_seed(seed)
    valid = torch.rand((R, SR), generator=g) < 0.6
    valid[:5] = False                      # empty rays
    valid[5:10] = True                     # full rays
    valid[10, :] = False
    valid[10, SR - 1] = True               # only the last slot: its delta clamps to 0
    o = torch.randn((R, 3), generator=g) * 0.1 + torch.tensor([2.0, 0.3, 0.2])
    d = torch.nn.functional.normalize(torch.randn((R, 3), generator=g), dim=-1)
    d[20] = torch.tensor([0.0, 0.0, 1.0])  # zero direction components: (p-o)/d = 0/0 -> NaN, skipped by nanmean
    zs = torch.sort(torch.rand((R, SR), generator=g) * 2.0 + 0.5, dim=1)[0]
    loc = (o[:, None] + zs[..., None] * d[:, None]) * valid[..., None]
    sdf = torch.randn((R, SR), generator=g) * 0.05
    sdf[30, :10] = 0.0                     # sign(0) = 0 edge
    sdf = torch.where(valid, sdf, torch.full_like(sdf, 1000.0))
    colors = torch.rand((R, SR, 3), generator=g) * valid[..., None]
    return valid, o, d, loc, sdf, colors


def _oracle_filter(valid, o, d, loc):
    t = ((loc - o.unsqueeze(1)) / d.unsqueeze(1)).nanmean(dim=-1)
    z = torch.where(valid, t, torch.zeros_like(t))
    zp = torch.cat([z, torch.zeros(z.shape[0], 1)], 1)
    deltas = torch.where(valid, zp[:, 1:] - zp[:, :-1], torch.zeros_like(z)).clamp(min=0)
    return z, deltas


def test_filter_points_matches_oracle():
    from spurfies_amd import ops

    valid, o, d, loc, _, _ = _inputs()
    z_o, dl_o = _oracle_filter(valid, o, d, loc)
    z, dl, x = ops.filter_points(loc.cuda(), valid.to(torch.uint8).cuda(), o.cuda(), d.cuda())
    np.testing.assert_allclose(z.cpu().numpy(), z_o.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), dl_o.numpy(), rtol=1e-5, atol=2e-6)
    x_o = o[:, None] + z_o[..., None] * d[:, None]
    np.testing.assert_allclose(x.view(*z.shape, 3).cpu().numpy(), x_o.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("SR", [80, 1, 130])
def test_render_forward_backward_match_oracle(SR):
    from spurfies_amd import ops

    valid, o, d, loc, sdf, colors = _inputs(R=200, SR=SR, seed=SR)
    z, deltas = _oracle_filter(valid, o, d, loc)
    beta_p = torch.tensor(0.07)
    coef = {k: torch.randn(s, generator=torch.Generator().manual_seed(1)) for k, s in
            dict(w=(200, SR), rgb=(200, 3), depth=(200, 1), dist=(200,), acc=(200, 1)).items()}

    def loss_of(weights, rgb, depth, dist, acc, dev):
        return sum((t * coef[k].to(dev)).sum() for k, t in dict(w=weights, rgb=rgb, depth=depth, dist=dist, acc=acc).items())

    # oracle (torch CPU autograd)
    s_o, c_o, b_o = sdf.clone().requires_grad_(True), colors.clone().requires_grad_(True), beta_p.clone().requires_grad_(True)
    beta_eff = b_o.abs() + 1e-4
    dens = torch.where(valid, P.laplace_density(s_o, beta_eff), torch.zeros_like(s_o))
    w_o = P.volume_weights(deltas, dens)
    rgb_o = (w_o.unsqueeze(-1) * c_o).sum(1)
    wsum = w_o.sum(-1, keepdim=True)
    depth_o = (w_o * z).sum(1, keepdim=True) / (wsum + 1e-8)
    dist_o = (w_o / (wsum + 1e-10) * z).sum(-1)
    loss_of(w_o, rgb_o, depth_o, dist_o, wsum, "cpu").backward()
    # HIP
    s_g, c_g, b_g = sdf.cuda().requires_grad_(True), colors.cuda().requires_grad_(True), beta_p.cuda().requires_grad_(True)
    out = ops.Render.apply(s_g, c_g, b_g.abs() + 1e-4, valid.to(torch.uint8).cuda(), z.cuda().contiguous(), deltas.cuda().contiguous())
    loss_of(*out, "cuda").backward()
    for got, want, nm in zip(out, (w_o, rgb_o, depth_o, dist_o, wsum), ("weights", "rgb", "depth", "dist", "acc")):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=nm)
    gs = s_o.grad.abs().max().item()
    np.testing.assert_allclose(s_g.grad.cpu().numpy(), s_o.grad.numpy(), rtol=2e-4, atol=2e-5 * gs, err_msg="g_sdf")
    np.testing.assert_allclose(c_g.grad.cpu().numpy(), c_o.grad.numpy(), rtol=2e-5, atol=2e-6, err_msg="g_colors")
    np.testing.assert_allclose(b_g.grad.item(), b_o.grad.item(), rtol=5e-4, err_msg="g_beta")


@pytest.mark.parametrize("SR", [80, 130])
def test_split_compositing_equals_the_fused_kernels_bit_for_bit(SR):
    """Round 5: the weights-only form of spf_render_forward + spf_render_rgb (and their backwards) — what a forked optimisation step uses so
    that the pseudo-point pass need not wait for the colour MLPs — give the SAME BITS as the fused kernels: same lane assignment, same
    reduction tree, the colour term enters the weight gradient as the same two-term sum."""
    from spurfies_amd import ops

    valid, o, d, loc, sdf, colors = _inputs(R=200, SR=SR, seed=3 + SR)
    z, deltas = _oracle_filter(valid, o, d, loc)
    gen = torch.Generator().manual_seed(2)
    coef = {k: torch.randn(s, generator=gen).cuda() for k, s in dict(rgb=(200, 3), depth=(200, 1), dist=(200,), acc=(200, 1), pts=(200, 3)).items()}
    sv, zc, dc = valid.to(torch.uint8).cuda(), z.cuda().contiguous(), deltas.cuda().contiguous()
    oc, dirc = o.cuda(), d.cuda()
    res = []
    for split in (False, True):
        s_g, c_g = sdf.cuda().requires_grad_(True), colors.cuda().requires_grad_(True)
        beta_p = torch.nn.Parameter(torch.tensor(0.07, device="cuda"))
        beta_p.grad = torch.zeros_like(beta_p)
        ops.set_grad_sinks([beta_p])
        beta = (beta_p.abs() + 1e-4).detach()
        if split:
            w, depth, dist, acc, pts = ops.RenderW.apply(s_g, beta, sv, zc, dc, beta_p, oc, dirc)
            rgb = ops.RenderRGB.apply(w, c_g)
        else:
            w, rgb, depth, dist, acc, pts = ops.Render.apply(s_g, c_g, beta, sv, zc, dc, beta_p, oc, dirc)
        loss = sum((t * coef[k]).sum() for k, t in dict(rgb=rgb, depth=depth, dist=dist, acc=acc, pts=pts).items())
        loss.backward()
        res.append([t.detach().clone() for t in (w, rgb, depth, dist, acc, pts, s_g.grad, c_g.grad, beta_p.grad)])
    for a, b, nm in zip(res[0], res[1], ("weights", "rgb", "depth", "dist", "acc", "pts", "g_sdf", "g_colors", "g_beta")):
        if nm == "g_beta":                      # one float atomic per ray: order noise only
            np.testing.assert_allclose(b.item(), a.item(), rtol=1e-5)
        else:
            assert torch.equal(a, b), nm
