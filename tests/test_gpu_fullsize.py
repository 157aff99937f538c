"""GPU parity AT BASELINE.json's sizes against the oracle (not HIP against HIP): the oracle runs one optimisation step on the same scene,
rays and CPU-generator draws and records its main-pass sample positions; the HIP path renders THOSE positions
(`PointVolSDF.render_points`: kNN, filter_points, SDF + normals, colours, compositing, pseudo-point / TV terms) and is held to the
oracle's losses (1e-4) and to the gradients of every trainable tensor (2e-3 of the tensor's gradient scale) — the sampler's last-bit
jitter stays out of the comparison, exactly as in tests/test_gpu_stages.py; a second check runs the full HIP forward (own sampler) at the
looser end-to-end bound.

  * configs[1]: 10^4 neural points, 1024-ray batch, fitted prior — the bench workload itself (oracle: ~10 s on 8 host threads);
  * configs[2] / configs[4] clouds (5*10^4 points in the +-2 grid; dense 2*10^5-point cloud): a 32-ray batch on the FULL cloud."""
import numpy as np
import pytest
import torch

from oracle import path as P
from spurfies_amd import synthetic as syn
from tests.helpers import assert_close_except_kinks
from tests.test_gpu_model import build_model

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
GRAD_RTOL = 2e-3


def oracle_step(scene, n_rays, view, seed):
    """-> (inputs, ground truth, oracle outputs, losses, grads, stages) of one oracle optimisation step."""
    torch.set_num_threads(min(8, torch.get_num_threads() if torch.get_num_threads() > 1 else 8))
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    g = torch.Generator().manual_seed(seed)
    uv = torch.from_numpy(syn.make_pixels(n_rays, g))[None]
    rgb_gt = torch.rand((n_rays, 3), generator=g)
    mask_gt = (torch.rand((n_rays,), generator=g) > 0.1).float()
    inp = {"intrinsics": torch.from_numpy(scene["intrinsics"])[None], "uv": uv, "pose": torch.from_numpy(scene["poses"][view])[None]}
    stages = {}
    torch.manual_seed(seed + 1)
    out, losses, grads = P.train_step_grads(inp, rgb_gt, mask_gt, st, cfg, stages=stages)
    return inp, rgb_gt, mask_gt, out, losses, grads, stages


def check_against_oracle(scene, n_rays, view, seed, min_points, weight_kink_frac=2e-4, weight_kink_rtol=None):
    from spurfies_amd import ops
    from spurfies_amd.model.loss import VolSDFLoss
    from spurfies_amd.train import TrainStep

    inp, rgb_gt, mask_gt, oout, olosses, ograds, stages = oracle_step(scene, n_rays, view, seed)
    assert int(stages["mask"].sum()) >= min_points, "the rays must actually hit the cloud"
    model = build_model(scene)
    model.keep_stages = True
    step = TrainStep(model)                       # default (reference-shaped) mode; owns the flat gradient buffer
    dev = {k: v.cuda() for k, v in inp.items()}
    dirs, loc, depth_scale = ops.camera_rays(dev["uv"], dev["pose"], dev["intrinsics"])
    out = model.render_points(stages["points"].detach().cuda(), dirs, loc, depth_scale, None)
    mask = stages["mask"].numpy()
    # the kNN of the oracle's sample positions: same slots, same neighbours (bit-exact)
    assert np.array_equal(model.stages["slot_valid"].bool().cpu().numpy(), mask)
    assert np.array_equal(model.stages["pidx"].view(*mask.shape, -1).cpu().numpy()[mask], stages["neighbor_idx"].numpy().astype(np.int32))
    sdf = model.stages["sdf"].detach().cpu().numpy()
    np.testing.assert_allclose(sdf[mask], stages["agg_sdf"].detach().numpy()[:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(model.stages["colors"].detach().cpu().numpy()[mask], stages["colors"].detach().numpy(), rtol=1e-4, atol=1e-5)
    for k in ("rgb_values", "weights", "depth_values"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), oout[k].detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    loss_fn = VolSDFLoss("torch.nn.L1Loss", local_weight=0.5, pseudo_weight=0.5, eikonal_weight=0.001, rgb_weight=1.0, tv_weight=0.01)
    gt = {"rgb": rgb_gt[None].cuda(), "mask": mask_gt[None, :, None].repeat(1, 1, 3).cuda()}
    losses = loss_fn(out, gt)
    for k, v in olosses.items():
        np.testing.assert_allclose(losses[k].item(), v.item(), rtol=LOSS_RTOL, atol=1e-6, err_msg=k)
    step.flat.zero_()
    losses["loss"].backward()
    names = {"neural_feats_color", "neural_feats_geometry", "density.beta"} | {n for n in ograds if n.startswith(("F_color", "R."))}
    assert names == set(ograds)
    for pname, p in model.named_parameters():
        if not p.requires_grad:
            continue
        want = ograds[pname].numpy()
        got = p.grad.detach().cpu().numpy()
        scale = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
        if pname.startswith("neural_feats"):
            # latent rows are fed by a handful of pairs each: one pair sitting on a LeakyReLU kink of the MLP behind it (a pre-activation
            # within rounding of 0 picks slope 1 or 0.01) moves ALL entries of its row.  Counted per ROW: at most 1.2 % of the rows that
            # received a gradient may hold an entry outside 2e-3 of the tensor's gradient scale, and such a row deviates by at most 5 % of its own
            # norm (measured: 0.07 - 2 %).  The allowance is what TRUE fp32 arithmetic needs against the CPU oracle's fp32: measured on MI355X on
            # the dense cloud (1121 colour rows touched), round 6 — the fp32-MFMA twin kernels (set_color_mode('f32')) 9 rows, the default H2
            # kernels 10, the bf16 x 3 kernels 2 (their products are exact to 2^-24, finer than an fp32 rounding); 1024 rays: 12 of 5536 (round 3).
            # It was 0.5 % while the bf16 x 3 kernels were the default.
            bad = ~np.isclose(got, want, rtol=GRAD_RTOL, atol=GRAD_RTOL * scale + 1e-12)
            bad_rows = bad.any(axis=1)
            touched = int((want != 0).any(axis=1).sum())
            if bad_rows.any():
                rn = np.linalg.norm(want[bad_rows], axis=1)
                dev = np.linalg.norm(got[bad_rows] - want[bad_rows], axis=1) / np.maximum(rn, 1e-30)
                print(f"{pname}: {int(bad_rows.sum())} of {touched} touched rows hold an entry outside {GRAD_RTOL}; their relative row deviation "
                      f"|got - want| / |want| = {np.sort(dev)[-5:]}, entries out per bad row {np.sort(bad[bad_rows].sum(1))[-5:]}, tensor rms {scale:.3e}")
            assert bad_rows.sum() <= max(1.2e-2 * touched, 2), (pname, int(bad_rows.sum()), touched)
            if bad_rows.any():
                assert float(dev.max()) <= 0.05, (pname, dev.max())
        else:
            assert_close_except_kinks(got, want, rtol=GRAD_RTOL, atol=GRAD_RTOL * scale + 1e-12, max_frac=weight_kink_frac, err_msg=pname)
            if weight_kink_rtol is not None:      # ... and what misses the tight bound misses it by little (a slope flip of ONE point's unit, spread over a dense gradient)
                np.testing.assert_allclose(got, want, rtol=weight_kink_rtol, atol=weight_kink_rtol * scale + 1e-12, err_msg=pname + " (kink bound)")
        np.testing.assert_allclose(float(p.grad.norm()), float(np.linalg.norm(want.astype(np.float64))), rtol=1e-3, err_msg=pname + " l2")
    return inp, gt, olosses, model


def test_configs1_bench_workload_full_step_matches_oracle():
    """BASELINE.json configs[1] — the exact shape bench.py times: 10^4 neural points, fitted prior, 1024 rays."""
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(10000, seed=0, prior="fitted")
    inp, gt, olosses, model = check_against_oracle(scene, 1024, view=0, seed=12345, min_points=30000)
    assert float(olosses["pseudo_loss"]) < 1.0, "fitted prior: the rendered points lie on the surface (no 1000 filler)"
    # the whole HIP step with its own sampler in the loop (sync-free mode, as the bench runs it): end-to-end bound
    model2 = build_model(scene)
    step = TrainStep(model2, sync_free=True)
    torch.manual_seed(12346)
    losses, _ = step({k: v.cuda() for k, v in inp.items()} | {"local_data": None}, gt)
    for k in ("loss", "rgb_loss", "mask_loss", "tv_loss", "pseudo_loss"):
        np.testing.assert_allclose(losses[k].item(), olosses[k].item(), rtol=2e-3, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(losses["eikonal_loss"].item(), olosses["eikonal_loss"].item(), rtol=1e-2, err_msg="eikonal (kink-borne)")


@pytest.mark.parametrize("n_points,spacing", [(50000, 0.025), (200000, 0.0125)])
def test_large_clouds_32_ray_batch_matches_oracle(n_points, spacing):
    """BASELINE configs[2] / configs[4] clouds at FULL size (garden-like 5*10^4 points in the +-2 grid; dense 2*10^5-point cloud: ~70 points
    per cell, far beyond upstream's 26-per-voxel cap), a 32-ray batch against the oracle."""
    scene = syn.make_scene(n_points, seed=21, spacing=spacing)
    assert tuple(scene["ranges"])[0] == -2.0
    scene["intrinsics"], scene["poses"] = syn.make_cameras(ring_radius=4.0)
    # Weight tensors: with 32 rays a single point's LeakyReLU kink in the HEAD (R.0 unit 122 of one dominant point, dense cloud) moves entries of every
    # dense weight gradient by 0.2 - 0.4 %: 15 - 121 entries per tensor (<= 0.4 %) miss the 2e-3 bound, none by more than 4.4e-3.  Measured on MI355X,
    # round 6: the fp32-MFMA twin of the head kernels (set_rhead_mode('f32')) and the default H2 head kernels give the SAME rows and counts to the
    # digit — they take the kink the way true fp32 arithmetic does — while the bf16 x 3 head kernels (products exact to 2^-24) side with the CPU
    # oracle and meet 2e-4 (the allowance while they were the default).  Allowed: 0.5 % of a tensor's entries outside 2e-3, every entry within 1e-2.
    check_against_oracle(scene, 32, view=0, seed=6, min_points=500, weight_kink_frac=5e-3, weight_kink_rtol=1e-2)
