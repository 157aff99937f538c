"""GPU: one optimisation-step forward + loss + backward against the oracle (CPU restatement, same CPU-generator draws) on the other
scene shapes BASELINE.json names — a cloud large enough for the +-2 grid (garden-like) and a dense cloud (spacing 0.0125, every
point has its k = 8 neighbours well inside the radius) — in the default and the sync-free form of the step."""
import numpy as np
import pytest
import torch

from oracle import path as P
from spurfies_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.mark.parametrize("sync_free", [False, True])
@pytest.mark.parametrize("n_points,spacing,want_range", [(26000, 0.025, 2.0), (9000, 0.0125, 1.0)])
def test_train_step_matches_oracle(n_points, spacing, want_range, sync_free):
    from spurfies_amd.conf import default_model_conf
    from spurfies_amd.model.pointneus_disent import PointVolSDF
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(n_points, seed=2, spacing=spacing)
    assert scene["ranges"][3] == want_range
    st = scene["state"]
    conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
    g = torch.Generator().manual_seed(11)
    n_rays = 96
    uv = torch.from_numpy(syn.make_pixels(n_rays, g))[None]
    K, pose = torch.from_numpy(scene["intrinsics"])[None], torch.from_numpy(scene["poses"][1])[None]
    rgb_gt, mask_gt = torch.rand((n_rays, 3), generator=g), (torch.rand((n_rays,), generator=g) > 0.1).float()
    step = TrainStep(model, sync_free=sync_free)
    torch.manual_seed(3)
    losses, out = step._forward_backward({"intrinsics": K.cuda(), "uv": uv.cuda(), "pose": pose.cuda(), "local_data": None},
                                         {"rgb": rgb_gt[None].cuda(), "mask": mask_gt[None, :, None].repeat(1, 1, 3).cuda()})
    ost = P.load_state(st)
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    torch.manual_seed(3)
    oout, olosses, ograds = P.train_step_grads({"intrinsics": K, "uv": uv, "pose": pose}, rgb_gt, mask_gt, ost, cfg)
    if not sync_free:                       # (the sync-free step keeps its counts on the device)
        assert model.stats["valid_points"] > 500
    np.testing.assert_allclose(out["rgb_values"].detach().cpu().numpy(), oout["rgb_values"].detach().numpy(), rtol=2e-3, atol=2e-4)
    for k in ("loss", "rgb_loss", "eikonal_loss"):
        assert float(losses[k].detach()) == pytest.approx(float(olosses[k].detach()), rel=2e-3, abs=1e-6), k
    params = dict(model.named_parameters())
    for name in ("density.beta", "R.4.weight", "R.0.weight", "F_color.0.weight", "F_color.4.bias", "neural_feats_color", "neural_feats_geometry"):
        g_dev = params[name].grad
        assert g_dev is not None, name
        assert _rel(g_dev, ograds[name]) < 2e-2, (name, _rel(g_dev, ograds[name]))      # summation order + isolated LeakyReLU kink flips
