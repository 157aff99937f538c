"""CPU: the oracle (oracle/path.py, oracle/voxel_grid.py) against fixtures produced by the
REFERENCE's own Python (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import path as P
from oracle.voxel_grid import VoxelGridOracle, brute_force_knn
from spurfies_amd import synthetic as syn
from tests.helpers import check_probes, draws_of, inputs_of, load_golden, scene_of

TOL = dict(rtol=2e-5, atol=2e-6)  # float32 round-off between two CPU evaluations of the same op sequence


TRAIN_FIXTURES = ["step_train_r128.npz", "step_train_far.npz", "step_train_garden.npz", "step_train_local.npz"]


@pytest.mark.parametrize("name", TRAIN_FIXTURES)
def test_train_step_matches_reference(name):
    """One reference optimisation step, stage by stage: default +-1 grid (r128, far), the +-2 grid the reference selects by scan
    name (garden), the fitted prior with `local_data` (find_surface_points + get_local_loss), then clip + Adam."""
    fx = load_golden(name)
    scene = scene_of(fx)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    if name == "step_train_garden.npz":
        assert cfg.ranges[0] == -2.0
    stages = {}
    out, losses, grads = P.train_step_grads(inputs_of(fx, scene), torch.from_numpy(fx["in.rgb_gt"]),
                                            torch.from_numpy(fx["in.mask_gt"]), st, cfg, draws=draws_of(fx), stages=stages)
    # integer / index stages are exact
    assert np.array_equal(stages["neighbor_idx"].numpy().astype(np.int32), fx["stage.neighbor_idx"])
    assert np.array_equal(stages["mask"].numpy(), fx["stage.mask"])
    assert np.array_equal(stages["ray_mask"].numpy(), fx["stage.ray_mask"])
    for k in ("points", "agg_sdf", "colors", "shading_pts", "z_slots", "deltas"):
        np.testing.assert_allclose(stages[k].detach().numpy(), fx[f"stage.{k}"], err_msg=k, **TOL)
    np.testing.assert_allclose(stages["z"].numpy(), fx["stage.z_vals"], **TOL)
    np.testing.assert_allclose(stages["grads"].detach().numpy(), fx["stage.gradients"], rtol=1e-4, atol=1e-5)
    if "stage.d_surface" in fx:
        # the reference calls find_surface_points with a leading batch dimension of 1 (pointneus_disent.py:738-743)
        assert np.array_equal(stages["network_mask"].numpy(), fx["stage.network_mask"][0]) and fx["stage.network_mask"].sum() > 30
        np.testing.assert_allclose(stages["d_surface"].detach().numpy(), fx["stage.d_surface"][0], **TOL)
        assert float(fx["out.local_loss"]) > 0
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz", "pseudo_pts_loss", "tv_loss", "grad_theta", "local_loss"):
        np.testing.assert_allclose(out[k].detach().numpy(), fx[f"out.{k}"], err_msg=k, **TOL)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), fx[f"loss.{k}"], rtol=2e-5, atol=1e-7, err_msg=k)
    for k, g in grads.items():
        check_probes(fx, f"grad.{k}", g, rtol=2e-4, atol=2e-7)
    # the step tail (train.py:359-363): clip_grad_norm_(1.0) + Adam(lr 5e-4) — parameter deltas of one update
    before = {k: v.detach().clone() for k, v in st.items() if v.requires_grad}
    opt, _ = P.make_optimizer(st)
    norm = P.optimizer_step(st, opt)
    np.testing.assert_allclose(float(norm), fx["adam.grad_norm"], rtol=2e-5)
    for k, b in before.items():
        delta = (st[k].detach() - b).reshape(-1).double()
        idx = torch.from_numpy(fx[f"adam.{k}.idx"])
        # Adam's first update is -lr * g / (|g| + eps): entries with |g| ~ eps = 1e-8 amplify round-off, all others agree tightly
        got, want = delta[idx].numpy(), fx[f"adam.{k}.val"]
        bad = ~np.isclose(got, want, rtol=1e-3, atol=5e-7)
        assert bad.mean() <= 0.02, (k, int(bad.sum()))


def test_sampler_matches_reference_g4():
    """the oracle's sampler alone against the reference's ErrorBoundSampler_pn driven by an analytic SDF."""
    fx = load_golden("sampler_g4.npz")
    dirs, cam = torch.from_numpy(fx["in.ray_dirs"]), torch.from_numpy(fx["in.cam_loc"])
    rad, wid = fx["meta.shell"]

    def shell(x):
        d = x.norm(dim=-1) - float(rad)
        return torch.where(d.abs() < float(wid), d, torch.full_like(d, 1000.0))

    cfg = P.PathConfig()
    st = {"density.beta": torch.tensor(0.1)}
    orig = P.sdf_at_points
    P.sdf_at_points = lambda x, grid, st_, cfg_: (shell(x), None)
    try:
        for tag, training, fast in (("train_fast1", True, 1), ("eval_full", False, -1), ("eval_fast1", False, 1)):
            draws = {k[len(tag) + 6:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(tag + ".draw.")}
            trace = {}
            z = P.error_bounded_z(dirs, cam, None, st, cfg, training, fast, draws, trace)
            assert trace["iters"] == len(fx[f"{tag}.calls"]), tag
            np.testing.assert_allclose(z.numpy(), fx[f"{tag}.z_vals"], rtol=2e-5, atol=2e-6, equal_nan=True, err_msg=tag)
    finally:
        P.sdf_at_points = orig
    assert len(fx["eval_full.calls"]) >= 2, "the eval case must iterate"


def test_voxel_thinning_matches_reference():
    """load_neural_points / voxelize / construct_vox_points_closest (spurfies/model/utils.py:6-88) — the
    product's torch_scatter-free restatement (runs on any device; here CPU) against the reference's output on a synthetic .ply."""
    import os
    import tempfile

    from spurfies_amd import synthetic as syn
    from spurfies_amd.model import utils as U
    from spurfies_amd.utils import surface

    fx = load_golden("voxelize.npz")
    pts, col = syn.make_raw_cloud(int(fx["meta.seed"]))
    assert float(pts.astype(np.float64).sum()) == float(fx["meta.checksum"])
    vox_res = int(fx["meta.vox_res"])
    centroid, grid_idx, min_idx = U.construct_vox_points_closest(torch.from_numpy(pts), vox_res)
    assert np.array_equal(grid_idx.numpy(), fx["out.grid_idx"])
    assert np.array_equal(min_idx.numpy(), fx["out.min_idx"])
    np.testing.assert_allclose(centroid.numpy(), fx["out.centroid"], rtol=1e-6, atol=1e-7)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cloud.ply")
        surface.write_ply_points(path, pts, col)
        res = U.load_neural_points(path, vox_res=vox_res, device="cpu")
    assert np.array_equal(res["pts"].numpy(), fx["out.pts"]) and np.array_equal(res["colors"].numpy(), fx["out.colors"])
    assert len(fx["out.pts"]) < len(pts) // 2, "the fixture must actually thin the cloud"


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


@pytest.mark.parametrize("name", ["trajectory_ref.npz", "trajectory_local_ref.npz", "trajectory_garden_ref.npz"])
def test_oracle_tracks_reference_trajectory(name):
    """G10 (tests/golden/trajectory_ref.npz: consecutive REFERENCE optimisation steps with one CPU-generator stream across them, Adam +
    clip + cosine schedule, train.py:330-364): the oracle, driven by its own optimiser restatement, stays on the reference's loss
    trajectory — i.e. it consumes the generator, updates every trainable and schedules the learning rate exactly like the reference."""
    fx = load_golden(name)
    scene = scene_of(fx)
    local = [None] * 3
    if bool(fx["meta.local"]):
        local = [{k: (torch.from_numpy(np.asarray(v)) if isinstance(v, (np.ndarray, np.floating)) else v)
                  for k, v in syn.make_local_data(scene, v_, seed=int(fx["meta.seed"])).items()} for v_ in range(3)]
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    grid = P.make_grid(cfg, st["neural_pts"])
    opt, sched = P.make_optimizer(st)
    K = torch.from_numpy(scene["intrinsics"])[None]
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    for i in range(8):
        inp = {"intrinsics": K, "uv": torch.from_numpy(fx["step.uv"][i])[None], "pose": torch.from_numpy(scene["poses"][int(fx["step.view"][i])])[None],
               "local_data": local[int(fx["step.view"][i])]}
        _, losses, _ = P.train_step_grads(inp, torch.from_numpy(fx["step.rgb_gt"][i]), torch.from_numpy(fx["step.mask_gt"][i]), st, cfg, grid=grid)
        norm = P.optimizer_step(st, opt, sched)
        for k, v in losses.items():
            np.testing.assert_allclose(v.item(), fx[f"loss.{k}"][i], rtol=2e-5, atol=1e-7, err_msg=f"step {i} {k}")
        np.testing.assert_allclose(float(norm), fx["step.grad_norm"][i], rtol=1e-4, err_msg=f"step {i} grad norm")
    np.testing.assert_allclose(float(P.get_beta(st, cfg).detach()), fx["step.beta"][7], rtol=1e-5)      # step.beta[i]: after step i's update


def test_eval_image_chunks_match_the_reference_in_its_evaluation_configuration():
    """eval_image_near0.npz: a 32 x 48 image rendered by the imported reference the way eval_spurfies.py:276-292 does (split_input chunks of
    500 pixels, fast = -1, merge_output) with the EVALUATION sampler range of config/confs/dtu_pn.conf:48 (near = 0.0).  The oracle renders
    the first full chunk and the short last one (36 pixels); the eval forward draws nothing that reaches the outputs."""
    fx = load_golden("eval_image_near0.npz")
    scene = scene_of(fx)
    st = P.load_state(scene["state"], requires_grad=False)
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]), near=float(fx["meta.near"]))
    assert cfg.near == 0.0
    grid = P.make_grid(cfg, st["neural_pts"])
    uv = torch.from_numpy(fx["in.uv"])
    total, chunk = uv.shape[0], int(fx["meta.chunk"])
    assert total == int(fx["meta.h"]) * int(fx["meta.w"]) and total % chunk != 0
    base = {"intrinsics": torch.from_numpy(scene["intrinsics"])[None], "pose": torch.from_numpy(scene["poses"][int(fx["meta.view"])])[None]}
    for lo in (0, (total // chunk) * chunk):
        hi = min(lo + chunk, total)
        stages = {}
        with torch.enable_grad():
            out = P.forward(dict(base, uv=uv[None, lo:hi]), st, cfg, grid=grid, training=False, fast=-1, stages=stages)
        for k in ("rgb_values", "depth_values", "normal_map", "weights"):
            np.testing.assert_allclose(out[k].detach().numpy().reshape(hi - lo, -1), fx[f"out.{k}"][lo:hi].reshape(hi - lo, -1), err_msg=f"{k} rows {lo}:{hi}",
                                       rtol=2e-4, atol=2e-5)
