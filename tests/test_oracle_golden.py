"""CPU: the oracle (oracle/path.py, oracle/voxel_grid.py) against fixtures produced by the
REFERENCE's own Python (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import path as P
from oracle.voxel_grid import VoxelGridOracle, brute_force_knn
from tests.helpers import check_probes, draws_of, inputs_of, load_golden, scene_of

TOL = dict(rtol=2e-5, atol=2e-6)  # float32 round-off between two CPU evaluations of the same op sequence


@pytest.mark.parametrize("name", ["step_train_r128.npz", "step_train_far.npz"])
def test_train_step_matches_reference(name):
    fx = load_golden(name)
    scene = scene_of(fx)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    stages = {}
    out, losses, grads = P.train_step_grads(inputs_of(fx, scene), torch.from_numpy(fx["in.rgb_gt"]),
                                            torch.from_numpy(fx["in.mask_gt"]), st, cfg, draws=draws_of(fx), stages=stages)
    # integer / index stages are exact
    assert np.array_equal(stages["neighbor_idx"].numpy().astype(np.int32), fx["stage.neighbor_idx"])
    assert np.array_equal(stages["mask"].numpy(), fx["stage.mask"])
    assert np.array_equal(stages["ray_mask"].numpy(), fx["stage.ray_mask"])
    np.testing.assert_allclose(stages["points"].numpy(), fx["stage.points"], **TOL)
    np.testing.assert_allclose(stages["agg_sdf"].detach().numpy(), fx["stage.agg_sdf"], **TOL)
    np.testing.assert_allclose(stages["colors"].detach().numpy(), fx["stage.colors"], **TOL)
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz", "pseudo_pts_loss", "tv_loss", "grad_theta"):
        np.testing.assert_allclose(out[k].detach().numpy(), fx[f"out.{k}"], err_msg=k, **TOL)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), fx[f"loss.{k}"], rtol=2e-5, atol=1e-7, err_msg=k)
    name_map = {"density.beta": "density.beta"}
    for k, g in grads.items():
        check_probes(fx, f"grad.{name_map.get(k, k)}", g, rtol=2e-4, atol=2e-7)


def test_eval_step_matches_reference():
    fx = load_golden("step_eval_r24.npz")
    scene = scene_of(fx)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    stages = {}
    out = P.forward(inputs_of(fx, scene), st, cfg, training=False, fast=-1, draws=draws_of(fx), stages=stages)
    assert stages["trace"]["iters"] == len(fx["meta.sampler_calls"])
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz", "normal_map", "pseudo_pts_loss", "tv_loss"):
        np.testing.assert_allclose(out[k].detach().numpy(), fx[f"out.{k}"], err_msg=k, rtol=5e-5, atol=5e-6)


def test_sdf_eval_matches_reference():
    fx = load_golden("sdf_eval_grid.npz")
    scene = scene_of(fx)
    st = P.load_state(scene["state"], requires_grad=False)
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    grid = P.make_grid(cfg, st["neural_pts"])
    with torch.no_grad():
        sdf, valid = P.sdf_at_points(torch.from_numpy(fx["in.x"]), grid, st, cfg)
    assert np.array_equal((fx["out.sdf"] != 1000.0), valid.numpy())
    np.testing.assert_allclose(sdf.numpy(), fx["out.sdf"], **TOL)


def test_knn_spec_fixture_and_bruteforce():
    fx = load_golden("knn_spec.npz")
    grid = VoxelGridOracle((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, (-1, -1, -1, 1, 1, 1))
    grid.set_pointset(fx["in.pts"])
    assert np.array_equal(grid.dims, fx["meta.dims"]) and np.array_equal(grid.origin, fx["meta.origin"])
    for nm in ("d1", "d98", "d128"):
        pidx, loc, slot_sample, ray_valid = grid.query_dense(fx[f"{nm}.x"], int(fx[f"{nm}.k"]), float(fx[f"{nm}.r"]), int(fx[f"{nm}.sr"]))
        assert np.array_equal(pidx, fx[f"{nm}.pidx"]) and np.array_equal(ray_valid, fx[f"{nm}.ray_valid"])
        assert np.array_equal(slot_sample, fx[f"{nm}.slot_sample"]) and np.array_equal(loc, fx[f"{nm}.loc"])
    # edge cases the fixture was built to hold
    assert (~fx["d98.ray_valid"][:8]).all(), "rays aimed away must be empty"
    assert (fx["d128.slot_sample"] >= 0).sum(1).max() == int(fx["d128.sr"]), "some ray must overflow SR"
    assert fx["d1.pidx"].max() < len(fx["in.pts"]) - 27, "out-of-range points must never be returned"
    # independent brute force on a subset
    x = fx["d1.x"][:300, 0]
    sel = fx["d1.slot_sample"][:300, 0] >= 0
    bf = brute_force_knn(fx["in.pts"], x[sel], 8, grid.radius(2), grid.ranges)
    assert np.array_equal(bf, fx["d1.pidx"][:300, 0][sel])


def test_eikonal_term_has_zero_gradient_for_trainables():
    """with LeakyReLU MLPs and detached RBF weights, d(eikonal)/d(latents) == 0, so the
    product path may drop the reference's double-backward (pointneus_disent.py:315-323)."""
    fx = load_golden("step_train_far.npz")
    scene = scene_of(fx)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    out = P.forward(inputs_of(fx, scene), st, cfg, training=True, fast=1, draws=draws_of(fx))
    eik = ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean()
    gs = torch.autograd.grad(eik, [st["neural_feats_geometry"], st["neural_feats_color"]], allow_unused=True)
    assert gs[1] is None or float(gs[1].abs().max()) == 0.0
    assert gs[0] is None or float(gs[0].abs().max()) == 0.0
