"""GPU: the feature-consistency ("local") term as HIP launches (csrc/local.hip).

Reference: find_surface_points (spurfies/model/pointneus_disent.py:586-612), the surface points (:744-749) and
feat_utils.get_local_loss (spurfies/feat_utils.py:377-451).  Checked against
  * the reference's own recorded stage tensors and loss / gradients (tests/golden/step_train_local.npz, made by oracle/make_golden.py),
    in the default mode (ops.LocalLoss), in the fused sync-free step (ops.LocalTerms through FusedLoss + Render) and as a replayed hipGraph;
  * the PyTorch formulation with the reference's function names (spurfies_amd/feat_utils.py: F.grid_sample + autograd) on random rows —
    channel counts above one wave's lanes, 1 - 3 source views, rays without a crossing, projections that leave the maps.
"""
import numpy as np
import pytest
import torch

from tests.helpers import check_probes_tight, inputs_of, load_golden, local_data_of, scene_of
from tests.test_gpu_stages import build_model

pytestmark = pytest.mark.gpu


def _random_rows(R, SR, seed, scene):
    """SDF / depth rows that look like a step's: increasing depths, a sign change on most rays, 1000-filled gaps, a few rays without one."""
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(0.9 + 2.6 * torch.rand((R, SR), generator=g), dim=-1).values
    t0 = 1.4 + 1.6 * torch.rand((R, 1), generator=g)                       # where the surface sits on each ray
    sdf = (t0 - z) * (0.5 + torch.rand((R, 1), generator=g)) + 0.02 * torch.randn((R, SR), generator=g)
    sdf[torch.rand((R, SR), generator=g) < 0.15] = 1000.0                   # slots without a neural point
    sdf[: R // 16] = sdf[: R // 16].abs().clamp(max=999.0)                  # rays that never cross
    sdf[R // 16: R // 8] = 1000.0                                           # rays without any point
    from spurfies_amd import synthetic as syn

    uv = torch.from_numpy(syn.make_pixels(R, g))[None].cuda()
    return sdf.cuda(), z.cuda(), uv


@pytest.mark.parametrize("C,view,n_src", [(8, 0, 2), (32, 1, 2), (70, 2, 2), (32, 0, 1), (16, 1, 3)])
def test_local_kernel_matches_the_pytorch_formulation(C, view, n_src):
    from spurfies_amd import feat_utils, ops
    from spurfies_amd import synthetic as syn
    from spurfies_amd.model.pointneus_disent import DistPoints, PointVolSDF

    scene = syn.make_scene(2000, seed=3)
    if n_src == 3:                                       # a fourth camera, so that a view has three sources
        K, poses = syn.make_cameras()
        scene = dict(scene, poses=np.concatenate([poses, poses[:1] @ np.diag([1.0, 1.0, 1.0, 1.0]).astype(poses.dtype)], 0))
        scene["poses"][3][:3, 3] *= 1.05
    local = syn.make_local_data(scene, view, channels=C, seed=11)
    local = {k: (torch.from_numpy(np.asarray(v)).cuda() if isinstance(v, (np.ndarray, np.floating)) else v) for k, v in local.items()}
    if n_src == 1:
        local["feat_src"], local["src_cams"] = local["feat_src"][:1].contiguous(), local["src_cams"][:1].contiguous()
    assert local["feat_src"].shape[0] == n_src
    R, SR = 512, 80
    sdf, z, uv = _random_rows(R, SR, seed=C + view, scene=scene)
    dirs, loc, _ = ops.camera_rays(uv, torch.from_numpy(scene["poses"][view])[None].cuda(), torch.from_numpy(scene["intrinsics"])[None].cuda())

    # PyTorch formulation (reference names) with autograd
    s_ref = sdf.clone().requires_grad_(True)
    d_ref, hit_ref = PointVolSDF.find_surface_points(s_ref, z)
    per_ref, cnt_ref = feat_utils.local_loss_terms(DistPoints.apply(loc, dirs, d_ref), hit_ref, local, per_point=True)
    (g_ref,) = torch.autograd.grad(per_ref.sum(), s_ref)

    # HIP
    s_hip = sdf.clone().requires_grad_(True)
    desc = feat_utils.local_desc(local, "cuda")
    tot, cnt, d_s, hit = ops.LocalLoss.apply(s_hip, z, loc, dirs, desc)
    (g_hip,) = torch.autograd.grad(tot, s_hip)
    per_hip = ops.local_forward(desc, sdf, z, loc, dirs).lsum

    assert int(hit_ref.sum()) > R // 2 and int((~hit_ref).sum()) >= R // 8
    assert torch.equal(hit, hit_ref)
    np.testing.assert_allclose(d_s.cpu().numpy(), d_ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert float(cnt) == float(cnt_ref) == n_src * int(hit_ref.sum())
    assert float(per_ref.detach().sum()) > 1.0                                        # the term is alive on these rows
    np.testing.assert_allclose(float(tot), float(per_hip.sum()), rtol=1e-6)
    # per ray: equal, except where a projection lands within rounding of the map border or |1 - cos| within rounding of 0.5 — the
    # reference's masks are step functions there (a ray of several hundred)
    same = (per_hip - per_ref.detach()).abs() <= 1e-5 + 1e-4 * per_ref.detach().abs()
    assert int((~same).sum()) <= max(1, R // 200), int((~same).sum())
    scale = float(g_ref.abs().max())
    assert scale > 0
    np.testing.assert_allclose(g_hip[same].cpu().numpy(), g_ref[same].cpu().numpy(), rtol=2e-3, atol=2e-4 * scale)
    assert (g_hip[~hit] == 0).all()
    assert int((g_hip != 0).sum()) <= 2 * int(hit.sum())                     # exactly the two slots of each crossing


def test_descriptor_is_cached_per_view_and_validates_shapes():
    from spurfies_amd import feat_utils
    from spurfies_amd import synthetic as syn

    scene = syn.make_scene(1500, seed=2)
    mk = lambda v: {k: (torch.from_numpy(np.asarray(x)).cuda() if isinstance(x, (np.ndarray, np.floating)) else x)
                    for k, x in syn.make_local_data(scene, v, channels=4, seed=1).items()}
    a, b = mk(0), mk(1)
    da = feat_utils.local_desc(a, "cuda")
    assert feat_utils.local_desc(a, "cuda") is da and feat_utils.local_desc(b, "cuda") is not da
    assert da.buf.dtype == torch.uint8 and da.buf.numel() == feat_utils.LocalDesc.NBYTES and da.n_src == 2
    a["cam"].mul_(1.0)                                                       # an in-place edit of a keyed tensor re-builds the descriptor
    assert feat_utils.local_desc(a, "cuda") is not da
    bad = dict(a, feat_src=a["feat_src"][:, :3])
    with pytest.raises(ValueError, match="feat_src"):
        feat_utils.local_desc(bad, "cuda")
    many = dict(a, feat_src=a["feat_src"].repeat(4, 1, 1, 1), src_cams=a["src_cams"].repeat(4, 1, 1, 1))
    with pytest.raises(ValueError, match="source views"):
        feat_utils.local_desc(many, "cuda")


@pytest.mark.parametrize("mode", ["sync_free", "graph"])
def test_fused_step_with_local_term_matches_the_reference_fixture(mode):
    """step_train_local.npz (the reference's forward + backward with local_data, fitted prior): the fused sync-free step — one launch for the
    term, its sum inside the loss kernels, its gradient inside the compositing backward — gives the reference's loss terms and gradients; so
    does the step replayed as a hipGraph that was CAPTURED ON ANOTHER VIEW (the replay reads this view's maps and cameras through the
    descriptor the step copies in front of it)."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    fx = load_golden("step_train_local.npz")
    scene = scene_of(fx)
    model = build_model(fx, scene)
    inp = inputs_of(fx, scene, device="cuda")
    gt = {"rgb": torch.from_numpy(fx["in.rgb_gt"])[None].cuda(), "mask": torch.from_numpy(fx["in.mask_gt"])[None, :, None].repeat(1, 1, 3).cuda()}
    step = TrainStep(model, sync_free=True, use_graph=(mode == "graph"), keep_grads=True)
    if mode == "graph":
        v2 = (int(fx["meta.view"]) + 1) % 3
        lo = syn.make_local_data(scene, v2, seed=int(fx["meta.seed"]))
        other = dict(inp, pose=torch.from_numpy(scene["poses"][v2])[None].cuda(),
                     local_data={k: (torch.from_numpy(np.asarray(v)).cuda() if isinstance(v, (np.ndarray, np.floating)) else v) for k, v in lo.items()})
        torch.manual_seed(1)
        with ops_owner(step):
            l_other, _ = step._graphed_forward_backward(other, gt)
            other_local = float(l_other["local_loss"])
        torch.manual_seed(int(fx["meta.seed"]) + 7)
        with ops_owner(step):
            losses, out = step._graphed_forward_backward(inp, gt)
        assert abs(float(losses["local_loss"]) - other_local) > 1e-4          # the two views' terms differ: the descriptor did change
    else:
        torch.manual_seed(int(fx["meta.seed"]) + 7)
        losses, out = step._forward_backward(inp, gt)
    assert float(fx["loss.local_loss"]) > 0.01
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), fx[f"loss.{k}"], rtol=2e-3 if k == "local_loss" else 2e-4, atol=2e-6, err_msg=k)
    for pname, p in model.named_parameters():
        if p.requires_grad:
            scale = float(fx[f"grad.{pname}.stats"][2]) / max(np.sqrt(p.numel()), 1.0)
            latent = pname.startswith("neural_feats")
            check_probes_tight(fx, f"grad.{pname}", p.grad, rtol=1e-3, atol=1e-3 * scale + 1e-9, outlier_frac=0.02 if latent else 0.0, outlier_rtol=5e-2)


def ops_owner(step):
    from spurfies_amd import ops

    return ops.scratch_owner(step)


def test_default_mode_stage_tensors_match_the_reference_fixture():
    """d_surface / network_mask as the kernel leaves them (model.stages) against the reference's recorded find_surface_points outputs, and
    the loss value, in the default (reference-shaped) training mode."""
    fx = load_golden("step_train_local.npz")
    scene = scene_of(fx)
    model = build_model(fx, scene)
    model.keep_stages = True
    model.train()
    from spurfies_amd import ops
    from spurfies_amd.train import TrainStep

    TrainStep(model)
    inp = inputs_of(fx, scene, device="cuda")
    _, _, depth_scale = ops.camera_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    out = model.render_points(torch.from_numpy(fx["stage.points"]).cuda(), torch.from_numpy(fx["stage.ray_dirs"]).cuda(),
                              torch.from_numpy(fx["stage.cam_loc"]).cuda(), depth_scale, local_data_of(fx, scene, "cuda"))
    ray_mask = fx["stage.ray_mask"]
    hit = model.stages["network_mask"].cpu().numpy()
    assert np.array_equal(hit[ray_mask], fx["stage.network_mask"][0]) and not hit[~ray_mask].any()
    np.testing.assert_allclose(model.stages["d_surface"].cpu().numpy()[ray_mask], fx["stage.d_surface"][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["local_loss"].item(), fx["out.local_loss"], rtol=1e-3)
