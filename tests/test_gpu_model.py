"""GPU end-to-end parity: PointVolSDF (HIP kNN + fused geometry kernels + host glue) against the
REFERENCE's recorded outputs (tests/golden/step_*.npz) and against the oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import check_probes_tight, assert_close_except_kinks, check_probes, draws_of, inputs_of, load_golden, scene_of

pytestmark = pytest.mark.gpu

# End-to-end tolerances (fp32): sample positions differ from the CPU run in the
# last bits (GPU transcendental / fma differences in the sampler), which moves SDF and colours by
# O(1e-5) and, rarely, flips a neighbour at the radius boundary.
# end-to-end outputs WITH the sampler in the loop.  Measured on MI355X (tools/e2e_errors.py, round 4): max |hip - reference| 5e-6 over rgb / depth /
# weights / xyz / normals of the three single-step fixtures (train fast=1 and the full evaluation loop) — the bound keeps a 4 x margin
OUT_TOL = dict(rtol=1e-4, atol=2e-5)


def build_model(scene, train=True, near=0.5):
    from spurfies_amd.conf import default_model_conf
    from spurfies_amd.model.pointneus_disent import PointVolSDF

    st = scene["state"]
    conf = default_model_conf(near=near, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model.freeze_prior()
    return model.train(train)


@pytest.mark.parametrize("name", ["step_train_r128.npz", "step_train_far.npz"])
def test_train_step_matches_reference_golden(name):
    from spurfies_amd.model.loss import VolSDFLoss

    fx = load_golden(name)
    scene = scene_of(fx)
    model = build_model(scene)
    inp = inputs_of(fx, scene, device="cuda")
    torch.manual_seed(int(fx["meta.seed"]) + 7)          # the generator state the reference run started from
    out = model(inp, fast=1)
    # the host consumed the CPU generator exactly like the reference did
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    assert np.array_equal(torch.rand((int(fx["meta.n_rays"]), 128)).numpy(), fx["draw.uniform_rand"])
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), fx[f"out.{k}"], err_msg=k, **OUT_TOL)
    np.testing.assert_allclose(out["tv_loss"].item(), fx["out.tv_loss"], rtol=1e-5)
    np.testing.assert_allclose(out["pseudo_pts_loss"].item(), fx["out.pseudo_pts_loss"], rtol=5e-4, atol=1e-6)
    # number of valid points may differ by a boundary flip or two, never by more
    assert abs(out["grad_theta"].shape[0] - fx["out.grad_theta"].shape[0]) <= 3
    if out["grad_theta"].shape == fx["out.grad_theta"].shape:
        assert_close_except_kinks(out["grad_theta"].detach().cpu().numpy(), fx["out.grad_theta"], rtol=5e-3, atol=5e-5, err_msg="grad_theta")
    loss_fn = VolSDFLoss("torch.nn.L1Loss", local_weight=0.5, pseudo_weight=0.5, eikonal_weight=0.001, rgb_weight=1.0, tv_weight=0.01)
    gt = {"rgb": torch.from_numpy(fx["in.rgb_gt"])[None], "mask": torch.from_numpy(fx["in.mask_gt"])[None, :, None].repeat(1, 1, 3)}
    losses = loss_fn(out, gt)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), fx[f"loss.{k}"], rtol=2e-4, atol=2e-6, err_msg=k)
    model.zero_grad()
    losses["loss"].backward()
    for pname, p in model.named_parameters():
        if p.requires_grad:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            scale = float(fx[f"grad.{pname}.stats"][2]) / max(np.sqrt(g.numel()), 1.0)
            latent = pname.startswith("neural_feats")
            check_probes_tight(fx, f"grad.{pname}", g, rtol=1e-3, atol=1e-3 * scale + 1e-9, outlier_frac=0.02 if latent else 0.0, outlier_rtol=5e-2)


@pytest.mark.parametrize("name", ["step_train_r128.npz", "step_train_far.npz"])
def test_sync_free_train_step_matches_reference_golden(name):
    """The bench's execution mode (no host synchronisation: worst-case buffers, device-side counts, fused camera / loss kernels,
    backward kernels adding straight into the flat gradient buffer) against the REFERENCE's recorded losses and gradients —
    including the fixture whose rays mostly miss the cloud."""
    from spurfies_amd.train import TrainStep

    fx = load_golden(name)
    scene = scene_of(fx)
    model = build_model(scene)
    step = TrainStep(model, sync_free=True)
    inp = inputs_of(fx, scene, device="cuda")
    gt = {"rgb": torch.from_numpy(fx["in.rgb_gt"])[None].cuda(), "mask": torch.from_numpy(fx["in.mask_gt"])[None, :, None].repeat(1, 1, 3).cuda()}
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    losses, out = step._forward_backward(inp, gt)                  # forward + loss + backward, no parameter update
    for k in ("rgb_values", "weights"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), fx[f"out.{k}"], err_msg=k, **OUT_TOL)
    for k, v in losses.items():
        np.testing.assert_allclose(v.item(), fx[f"loss.{k}"], rtol=2e-4, atol=2e-6, err_msg=k)
    for pname, p in model.named_parameters():
        if p.requires_grad:
            scale = float(fx[f"grad.{pname}.stats"][2]) / max(np.sqrt(p.numel()), 1.0)
            latent = pname.startswith("neural_feats")
            check_probes_tight(fx, f"grad.{pname}", p.grad, rtol=1e-3, atol=1e-3 * scale + 1e-9, outlier_frac=0.02 if latent else 0.0, outlier_rtol=5e-2)


def test_eval_step_matches_reference_golden():
    fx = load_golden("step_eval_r24.npz")
    scene = scene_of(fx)
    model = build_model(scene, train=False)
    inp = inputs_of(fx, scene, device="cuda")
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    out = model(inp, fast=-1)
    assert model.ray_sampler.last_iters == len(fx["meta.sampler_calls"])
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz", "normal_map"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), fx[f"out.{k}"], err_msg=k, **OUT_TOL)      # measured: 5e-6 (tools/e2e_errors.py)


def test_eval_keys_give_the_full_forwards_render_outputs():
    """PointVolSDF.eval_keys (what the streamed evaluation renderers ask for: the outputs the reference's evaluation loops read,
    train.py:419-424 / eval_spurfies.py:282-287) against the full reference-shaped evaluation forward on the same rays: rgb, depth and
    weights bit for bit (same kernels, same order), normals to round-off (composited inside the render launch instead of by torch ops) —
    and against the reference fixture itself."""
    fx = load_golden("step_eval_r24.npz")
    scene = scene_of(fx)
    model = build_model(scene, train=False)
    inp = inputs_of(fx, scene, device="cuda")
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    with torch.no_grad():
        full = model(inp, fast=-1)
    model.eval_keys = ("rgb_values", "depth_values", "normal_map", "weights")
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    with torch.no_grad():
        light = model(inp, fast=-1)
    model.eval_keys = None
    assert set(light) == {"rgb_values", "depth_values", "normal_map", "weights"}
    for k in ("rgb_values", "depth_values", "weights"):
        assert torch.equal(light[k].reshape(full[k].shape), full[k]), k
    np.testing.assert_allclose(light["normal_map"].cpu().numpy(), full["normal_map"].cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in light:
        np.testing.assert_allclose(light[k].cpu().numpy().reshape(fx[f"out.{k}"].shape), fx[f"out.{k}"], err_msg=k, **OUT_TOL)


def test_sdf_eval_matches_reference_golden():
    fx = load_golden("sdf_eval_grid.npz")
    scene = scene_of(fx)
    model = build_model(scene, train=False)
    with torch.no_grad():
        sdf = model.get_sdf_eval(torch.from_numpy(fx["in.x"]).cuda())
    got = sdf.cpu().numpy()
    assert np.array_equal(got != 1000.0, fx["out.sdf"] != 1000.0)
    np.testing.assert_allclose(got, fx["out.sdf"], rtol=1e-4, atol=5e-6)


def test_state_dict_keys_match_reference_contract():
    """SURVEY.md §5 checkpoint contract: reference checkpoints must load."""
    from spurfies_amd import synthetic as syn

    scene = syn.make_scene(500, seed=0)
    model = build_model(scene)
    want = {"neural_pts", "neural_feats_color", "neural_feats_geometry", "density.beta"}
    want |= {f"F_color.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")}
    want |= {f"F_geometry.{i}.{n}" for i in (0, 2, 4, 6, 8) for n in ("weight", "bias")}
    want |= {f"T.0.{n}" for n in ("weight", "bias")} | {f"R.{i}.{n}" for i in (0, 2, 4) for n in ("weight", "bias")}
    assert set(model.state_dict().keys()) == want


def test_three_optimisation_steps_track_the_oracle():
    """SURVEY.md §8(c) G6 (+ Adam): three full steps (forward, loss, backward, clip 1.0, Adam lr 5e-4, cosine schedule) on
    the GPU stay on the trajectory of the oracle driven by the same optimiser recipe on the CPU."""
    from oracle import path as P
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(2500, seed=9)
    model = build_model(scene)
    step = TrainStep(model)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    grid = P.make_grid(cfg, st["neural_pts"])
    names = [k for k, v in st.items() if v.requires_grad]
    opt = torch.optim.Adam([{"params": [], "lr": 1e-2}, {"params": [st[k] for k in names], "lr": 5e-4}])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100_000, eta_min=3e-4)
    g = torch.Generator().manual_seed(77)
    K = torch.from_numpy(scene["intrinsics"])[None]
    for it in range(3):
        uv = torch.from_numpy(syn.make_pixels(48, g))[None]
        pose = torch.from_numpy(scene["poses"][it % 3])[None]
        rgb, mask = torch.rand((48, 3), generator=g), (torch.rand((48,), generator=g) > 0.2).float()
        torch.manual_seed(100 + it)
        losses, _ = step({"intrinsics": K.cuda(), "uv": uv.cuda(), "pose": pose.cuda(), "local_data": None},
                         {"rgb": rgb[None].cuda(), "mask": mask[None, :, None].repeat(1, 1, 3).cuda()})
        torch.manual_seed(100 + it)
        _, olosses, _ = P.train_step_grads({"intrinsics": K, "uv": uv, "pose": pose}, rgb, mask, st, cfg, grid=grid)
        torch.nn.utils.clip_grad_norm_([st[k] for k in names], 1.0)
        opt.step()
        sched.step()
        np.testing.assert_allclose(losses["loss"].item(), olosses["loss"].item(), rtol=3e-3, err_msg=f"step {it}")
    sd = model.state_dict()
    for k in names:
        a, b = sd[k].detach().cpu().numpy(), st[k].detach().numpy()
        moved = np.abs(b - np.asarray(scene["state"][k])).max()
        assert moved > 0, k
        # after 3 Adam steps every touched entry moved by ~3*lr; the two trajectories must agree far below that
        np.testing.assert_allclose(a, b, rtol=0, atol=0.1 * moved + 1e-7, err_msg=k)


@pytest.mark.parametrize("name", ["trajectory_ref.npz", "trajectory_local_ref.npz", "trajectory_garden_ref.npz"])
def test_multi_step_trajectory_tracks_the_reference(name):
    """G10: 200 consecutive optimisation steps of the REFERENCE itself (tests/golden/trajectory_ref.npz, generated by
    oracle/make_golden.py from the imported reference: forward fast=1, VolSDFLoss, backward, clip 1.0, Adam, cosine schedule, the three
    views in turn, ONE CPU-generator stream across all steps) replayed through the sync-free HIP step.  A step's random draws continue
    where the previous step stopped, so a single extra or missing draw would derail the run at step 2.  fp32 differences are amplified
    by Adam from step to step; measured on MI355X: total loss within 2.3e-7 over the first 10 steps, 1.1e-4 over the first 50, 2.3 %
    at worst over 200; norms of the parameters' total change within 1.4 % (latents) to 5.6 % (a bias vector); the final get_sdf_eval volume
    within 0.7 % of its range of the reference's, zero level sets 0.0006 grid steps apart (Chamfer).
    trajectory_local_ref.npz: 60 steps with `local_data` on every step (find_surface_points + get_local_loss at local_weight 0.5)."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    fx = load_golden(name)
    scene = scene_of(fx)
    local = [None] * 3
    if bool(fx["meta.local"]):
        local = [{k: (torch.from_numpy(np.asarray(v)).cuda() if isinstance(v, (np.ndarray, np.floating)) else v)
                  for k, v in syn.make_local_data(scene, v_, seed=int(fx["meta.seed"])).items()} for v_ in range(3)]
    model = build_model(scene)
    step = TrainStep(model, sync_free=True)
    before = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    K = torch.from_numpy(scene["intrinsics"])[None].cuda()
    n = int(fx["meta.steps"])
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    got = []
    for i in range(n):
        inp = {"intrinsics": K, "uv": torch.from_numpy(fx["step.uv"][i])[None].cuda(),
               "pose": torch.from_numpy(scene["poses"][int(fx["step.view"][i])])[None].cuda(), "local_data": local[int(fx["step.view"][i])], "iter_step": i}
        gt = {"rgb": torch.from_numpy(fx["step.rgb_gt"][i])[None].cuda(), "mask": torch.from_numpy(fx["step.mask_gt"][i])[None, :, None].repeat(1, 1, 3).cuda()}
        losses, _ = step(inp, gt)
        got.append(losses)                       # device scalars: no host read-back inside the loop
    for k in ("loss", "rgb_loss", "tv_loss", "mask_loss", "pseudo_loss", "eikonal_loss") + (("local_loss",) if bool(fx["meta.local"]) else ()):
        ref, mine = fx[f"loss.{k}"], np.asarray([g[k].item() for g in got])
        # the eikonal and feature-consistency terms sit on kinks (a neighbour set / a zero crossing flipping between two samples)
        kinky = k in ("eikonal_loss", "local_loss")
        np.testing.assert_allclose(mine[:10], ref[:10], rtol=5e-3 if kinky else 1e-4, err_msg=k)
        np.testing.assert_allclose(mine[:50], ref[:50], rtol=3e-2 if kinky else 5e-3, err_msg=k)
        np.testing.assert_allclose(mine, ref, rtol=0.15, err_msg=k)
        np.testing.assert_allclose(mine[-20:].mean(), ref[-20:].mean(), rtol=0.03, err_msg=k)
    state = step.optimizer._flat["state"].tolist()        # {t, skipped (non-finite) updates, last gradient norm, last clip coefficient}
    assert int(state[0]) == n and int(state[1]) == 0, state
    for pname, p in model.named_parameters():
        if p.requires_grad:
            norm = float((p.detach() - before[pname]).double().norm())
            ref_norm = float(fx[f"delta.{pname}.stats"][2])
            print(f"{name}: |delta {pname}| {norm:.6e} vs reference {ref_norm:.6e} ({(norm / ref_norm - 1) * 100:+.2f} %)")
            # measured on MI355X over 200 steps: latents within 1.4 %, weights within 4 %, the two most drifting bias vectors -5.6 % / -5.1 %;
            # the 60- and 30-step runs within 0.05 %
            # Bias vectors drift most: over four MI355X runs of the 200-step fixture F_color.0.bias / F_color.2.bias landed between -5 % and
            # -11 %, the 3-element R.4.bias between -1.4 % and -11 %: 20 % for 1-D tensors, 10 % for weight matrices, 5 % for the latent tables.
            # Round 4: this is the TRAJECTORY's sensitivity, not the kernels' — the CPU oracle (bit-equal to the reference's losses for the first
            # 10 steps) ends -2 ... -15 % off the fixture on the same tensors when only its thread count changes
            # (tools/traj_sensitivity.py -> profiles/r04_traj_sensitivity.json), and the bit-reproducible HIP mode lands at one fixed
            # number per tensor inside that spread (tools/traj_fixed.py -> profiles/r04_traj_fixed.json).
            tol = 0.05 if pname.startswith("neural_feats") else (0.20 if p.dim() <= 1 else 0.10)
            np.testing.assert_allclose(norm, ref_norm, rtol=tol, err_msg=pname)
    np.testing.assert_allclose(float(model.density.get_beta().detach()), fx["step.beta"][-1], rtol=0.03)
    # ---- "at equal Chamfer" (BASELINE.json): the geometry the optimisation ENDS with.  The fixture holds the reference model's get_sdf_eval
    # volume after its last step, over the reference's evaluation grid (plots.py:302-333); the HIP-trained model is swept over the same grid.
    from spurfies_amd.utils import surface

    model.eval()
    grid = surface.get_grid(torch.from_numpy(scene["state"]["neural_pts"]), int(fx["final.grid_resolution"]), eps=float(fx["final.grid_eps"]))
    vol = surface.sdf_volume(model.get_sdf_eval, grid)
    ref = fx["final.sdf_volume"]
    assert vol.shape == ref.shape
    defined = ref != 1000.0
    assert np.array_equal(vol != 1000.0, defined)                       # the cloud is static: same support, bit for bit
    d = np.abs(vol[defined] - ref[defined])
    span = float(np.abs(ref[defined]).max())
    h = float(grid["xyz"][0][1] - grid["xyz"][0][0])
    pts_h, pts_r = surface.surface_points(vol, grid), surface.surface_points(ref, grid)
    assert len(pts_r) > 500 and abs(len(pts_h) - len(pts_r)) <= 0.02 * len(pts_r)
    cd, acc, comp = surface.chamfer(pts_h, pts_r)
    print(f"{name}: final SDF after {n} steps: max |hip - ref| {d.max():.3e} (span {span:.3f}), mean {d.mean():.3e}; zero level sets: "
          f"{len(pts_h)} vs {len(pts_r)} points, Chamfer {cd / h:.4f} grid steps (accuracy {acc / h:.4f}, completeness {comp / h:.4f})")
    # SDF values agree to 2 % of the value + 0.2 % of the volume's range (fp32 drift through n Adam steps; kinks move single samples)
    assert (d <= 0.02 * np.abs(ref[defined]) + 2e-3 * span).mean() >= 0.999
    assert cd < 0.05 * h, cd / h                                         # the two surfaces coincide far below a grid step (bar: 0.3)
    # ... and the comparison is not vacuous: against the geometry the run STARTED from, the optimisation moved the field by far more than
    # the two implementations differ at the end
    fresh = build_model(scene, train=False)
    vol0 = surface.sdf_volume(fresh.get_sdf_eval, grid)
    moved = float(np.abs(vol0[defined] - ref[defined]).mean())
    print(f"{name}: mean |initial - reference final| {moved:.3e} vs mean |hip final - reference final| {d.mean():.3e}")
    assert d.mean() < 0.2 * moved


@pytest.mark.parametrize("n_rays", [96, 125, 126])
def test_sync_free_step_equals_default_step(n_rays):
    """The static-shape / device-side-count path (no host synchronisation) reproduces the default step: same loss, same
    gradients (up to float-atomic order) and the same CPU-generator consumption.  125 / 126 rays: R * SR is not a multiple of 64 — the
    narrow row-major problem of the batched weight-gradient launch takes any row count (round-5 advisor finding: --rays 1000 --gpus 8)."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(3000, seed=12)
    g = torch.Generator().manual_seed(3)
    uv = torch.from_numpy(syn.make_pixels(n_rays, g))[None].cuda()
    K, pose = torch.from_numpy(scene["intrinsics"])[None].cuda(), torch.from_numpy(scene["poses"][2])[None].cuda()
    gt = {"rgb": torch.rand((n_rays, 3), generator=g)[None].cuda(), "mask": (torch.rand((n_rays,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}
    res = []
    for sync_free in (False, True):
        model = build_model(scene)
        step = TrainStep(model, sync_free=sync_free, keep_grads=True)       # the test reads the gradient buffer after the step
        torch.manual_seed(5)
        losses, out = step({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None}, gt)
        after = torch.rand(1).item()              # generator state after the step
        res.append((losses, step.flat.buffer.clone(), after, out))
    (l0, g0, a0, o0), (l1, g1, a1, o1) = res
    assert a0 == a1
    for k in l0:
        np.testing.assert_allclose(l1[k].item(), l0[k].item(), rtol=1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(o1["rgb_values"].detach().cpu().numpy(), o0["rgb_values"].detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=2e-3, atol=2e-5 * float(g0.abs().max()))


def test_sync_free_steps_run_ahead_without_tearing_draws():
    """Sync-free steps never synchronise, so the host runs several steps ahead of the GPU while the CPU-generator draws travel
    through pinned staging buffers: with a ring of only two buffers and eight steps queued back to back, every step must still
    consume exactly its own draws — the loss trajectory equals the default (synchronising) mode's."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(3000, seed=14)
    g = torch.Generator().manual_seed(8)
    K = torch.from_numpy(scene["intrinsics"])[None].cuda()
    batches = []
    for it in range(8):
        uv = torch.from_numpy(syn.make_pixels(256, g))[None].cuda()
        pose = torch.from_numpy(scene["poses"][it % 3])[None].cuda()
        gt = {"rgb": torch.rand((256, 3), generator=g)[None].cuda(), "mask": (torch.rand((256,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}
        batches.append(({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None}, gt))
    traj = []
    for sync_free in (False, True):
        model = build_model(scene)
        step = TrainStep(model, sync_free=sync_free)
        step.N_STAGING = 2
        torch.manual_seed(21)
        torch.cuda.synchronize()
        ls = [step(*b)[0]["loss"] for b in batches]           # device scalars: nothing is read back between the steps
        traj.append([float(v.item()) for v in ls])
    np.testing.assert_allclose(traj[1], traj[0], rtol=5e-4)


def test_fixed_scatter_mode_makes_the_whole_step_bit_reproducible():
    """ops.set_scatter_mode('fixed') — every atomically summed quantity of a step becomes order-independent: the three latent-gradient
    scatters (colour backward, geometry backward, TV), the forward's RBF-weighted mean, R.4's weight / bias and beta accumulate 2^-48
    fixed-point integers with 64-bit integer atomics; the weight-gradient GEMMs reduce their partial slabs and column sums in block order
    (SPF_WGRAD_DETERMINISTIC).  Two runs of the same step then agree BIT FOR BIT in every loss term and in the gradient of EVERY trainable
    tensor — and in the parameters after three optimisation steps; against the float-atomic default they differ by summation noise only."""
    from spurfies_amd import ops
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(4000, seed=15, prior="fitted")
    g = torch.Generator().manual_seed(5)
    K = torch.from_numpy(scene["intrinsics"])[None].cuda()
    batches = []
    for it in range(3):
        uv = torch.from_numpy(syn.make_pixels(512, g))[None].cuda()
        pose = torch.from_numpy(scene["poses"][it % 3])[None].cuda()
        batches.append(({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None},
                        {"rgb": torch.rand((512, 3), generator=g)[None].cuda(), "mask": (torch.rand((512,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}))
    runs = {}
    try:
        for mode in ("fixed", "fixed", "atomic"):
            ops.set_scatter_mode(mode)
            model = build_model(scene)
            step = TrainStep(model, sync_free=True, keep_grads=True)
            torch.manual_seed(31)
            losses, _ = step._forward_backward(dict(batches[0][0]), batches[0][1])
            grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
            terms = {k: v.detach().clone() for k, v in losses.items()}
            torch.manual_seed(31)
            for b in batches:                                 # three full optimisation steps (clip + Adam included)
                step(dict(b[0]), b[1])
            params = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
            runs.setdefault(mode, []).append((grads, terms, params))
    finally:
        ops.set_scatter_mode("atomic")
    (g0, t0, p0), (g1, t1, p1) = runs["fixed"]
    assert set(g0) >= {"neural_feats_color", "neural_feats_geometry", "F_color.0.weight", "F_color.0.bias", "F_color.4.weight", "F_color.6.weight",
                       "R.0.weight", "R.0.bias", "R.2.weight", "R.4.weight", "R.4.bias", "density.beta"}
    for n in g0:
        assert float(g0[n].abs().max()) > 0, n
        assert torch.equal(g0[n], g1[n]), f"gradient of {n} must be bit-reproducible in scatter mode 'fixed'"
        assert torch.equal(p0[n], p1[n]), f"{n} after three optimisation steps must be bit-reproducible in scatter mode 'fixed'"
    for k in t0:
        assert torch.equal(t0[k], t1[k]), k
    ga = runs["atomic"][0][0]
    for n in g0:                                              # same sums as the float-atomic default, up to its summation noise
        a, b = g0[n].cpu().numpy(), ga[n].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6 * float(np.abs(b).max()), err_msg=n)


@pytest.mark.parametrize("graph", [False, True])
def test_resumed_run_continues_bit_for_bit_in_fixed_scatter_mode(tmp_path, graph):
    """TrainStep.state_dict() (model, Adam moments, cosine schedule, step counter, the CPU generator of the reference's per-step draws —
    ray_sampler.py:55,514,550,562) written after three steps and loaded into a REBUILT model + step: the next three steps end in exactly
    the parameters and moments of the uninterrupted six-step run (ops.set_scatter_mode("fixed"): no summation-order noise)."""
    from spurfies_amd import ops
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(4000, seed=15, prior="fitted")
    g = torch.Generator().manual_seed(6)
    K = torch.from_numpy(scene["intrinsics"])[None].cuda()
    batches = []
    for it in range(6):
        uv = torch.from_numpy(syn.make_pixels(256, g))[None].cuda()
        pose = torch.from_numpy(scene["poses"][it % 3])[None].cuda()
        batches.append(({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None},
                        {"rgb": torch.rand((256, 3), generator=g)[None].cuda(), "mask": (torch.rand((256,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}))

    def flat_state(step):
        f = step.optimizer._flat
        return [f[k].clone() for k in ("param", "m", "v")] + [torch.tensor(step.optimizer.param_groups[1]["lr"])]

    ops.set_scatter_mode("fixed")
    try:
        step = TrainStep(build_model(scene), sync_free=True, use_graph=graph)
        torch.manual_seed(3)
        for i, b in enumerate(batches):
            step(dict(b[0]), b[1])
            if i == 1:
                # a device-wide synchronisation between two replays of one captured step: with a hipMemsetAsync node in the graph (the
                # accumulators' status word used to be cleared that way) every replay after it flushed NaN on ROCm 7.2 and Adam skipped
                torch.cuda.synchronize()
        assert step.optimizer._flat["state"].tolist()[:2] == [6.0, 0.0], "six updates, none skipped"
        straight = flat_state(step)
        step = TrainStep(build_model(scene), sync_free=True, use_graph=graph)
        torch.manual_seed(3)
        for b in batches[:3]:
            step(dict(b[0]), b[1])
        torch.save(step.state_dict(), tmp_path / "state.pth")
        del step
        torch.manual_seed(999)                                   # a resumed process starts from some other generator state
        step = TrainStep(build_model(scene), sync_free=True, use_graph=graph)
        step.load_state_dict(torch.load(tmp_path / "state.pth", map_location="cuda"))
        assert step.iter_step == 3 and step.scheduler.last_epoch == 3
        for b in batches[3:]:
            step(dict(b[0]), b[1])
        resumed = flat_state(step)
    finally:
        ops.set_scatter_mode("atomic")
    for name, a, b in zip(("parameters", "exp_avg", "exp_avg_sq", "lr"), straight, resumed):
        assert torch.equal(a.cpu(), b.cpu()), f"{name} of the resumed run differ from the uninterrupted run"


def test_graphed_step_tracks_eager_step():
    """hipGraph replay of forward + loss + backward gives the same three-step trajectory as eager launches."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(3000, seed=13)
    g = torch.Generator().manual_seed(4)
    K = torch.from_numpy(scene["intrinsics"])[None].cuda()
    batches = []
    for it in range(3):
        uv = torch.from_numpy(syn.make_pixels(64, g))[None].cuda()
        pose = torch.from_numpy(scene["poses"][it])[None].cuda()
        gt = {"rgb": torch.rand((64, 3), generator=g)[None].cuda(), "mask": (torch.rand((64,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}
        batches.append(({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None}, gt))
    traj = []
    for use_graph in (False, True):
        model = build_model(scene)
        step = TrainStep(model, sync_free=True, use_graph=use_graph)
        torch.manual_seed(9)
        ls = [step(*b)[0]["loss"].item() for b in batches]
        traj.append((ls, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = traj
    np.testing.assert_allclose(l1, l0, rtol=2e-5)
    for k in s0:
        np.testing.assert_allclose(s1[k].cpu().numpy(), s0[k].cpu().numpy(), rtol=0, atol=2e-5, err_msg=k)


def test_forked_step_equals_the_single_stream_step():
    """Round 5: TrainStep(fork=True) issues independent passes on side streams (parallel branches of the captured graph): weight packing + TV
    beside the geometry kernel, the pseudo-point pass beside the colour stage (split compositing), the head's weight-gradient GEMMs and the
    geometry passes' latent scatters beside the colour backward.  Same kernels, same sums: losses, rendered colours and the whole gradient
    buffer agree with the single-stream step (float-atomic order noise only), eagerly and as a graph; the CPU generator advances alike."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(3000, seed=17, prior="fitted")
    g = torch.Generator().manual_seed(6)
    uv = torch.from_numpy(syn.make_pixels(128, g))[None].cuda()
    K, pose = torch.from_numpy(scene["intrinsics"])[None].cuda(), torch.from_numpy(scene["poses"][1])[None].cuda()
    gt = {"rgb": torch.rand((128, 3), generator=g)[None].cuda(), "mask": (torch.rand((128,), generator=g) > 0.2).float()[None, :, None].repeat(1, 1, 3).cuda()}
    res = {}
    for name, kw in (("plain", dict(sync_free=True, fork=False)), ("fork", dict(sync_free=True, fork=True)),
                     ("graph", dict(use_graph=True)), ("graph_fork", dict(use_graph=True, fork=True))):
        model = build_model(scene)
        step = TrainStep(model, keep_grads=True, **kw)
        assert step.fork == (name in ("fork", "graph_fork"))
        torch.manual_seed(5)
        for _ in range(3 if "graph" in name else 1):          # a graph replays the capture's buffers: three replays, same inputs
            torch.manual_seed(5)
            losses, out = step({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None}, gt)
        after = torch.rand(1).item()
        torch.cuda.synchronize()
        res[name] = ({k: float(v.item()) for k, v in losses.items()}, step.flat.buffer.clone(), after, out["rgb_values"].detach().clone())
    l0, g0, a0, rgb0 = res["plain"]
    assert float(g0.abs().max()) > 0
    for name in ("fork", "graph_fork"):
        l1, g1, a1, rgb1 = res[name]
        base = res["plain"] if name == "fork" else res["graph"]
        assert a1 == base[2], name
        for k in l0:
            np.testing.assert_allclose(l1[k], base[0][k], rtol=1e-5, atol=1e-7, err_msg=f"{name}: {k}")
        np.testing.assert_allclose(rgb1.cpu().numpy(), base[3].cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(g1.cpu().numpy(), base[1].cpu().numpy(), rtol=2e-3, atol=2e-5 * float(g0.abs().max()), err_msg=name)


def test_two_forked_graphs_replayed_on_two_streams_do_not_share_scratch():
    """Two scenes with graph-replayed, forked steps on two streams (MultiSceneTrainer; configs[3] at strong-scaled batch sizes): every
    workspace (compaction words, weight-gradient slabs, loss partials) belongs to its step, so each scene's three-step trajectory is the one
    it has when trained alone (round-4 advisor finding: stream-keyed caches were baked into every captured graph)."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import MultiSceneTrainer, TrainStep

    def make(seed):
        scene = syn.make_scene(2500, seed=seed, prior="fitted")
        model = build_model(scene)
        g = torch.Generator().manual_seed(seed + 50)
        K = torch.from_numpy(scene["intrinsics"])[None].cuda()
        bs = []
        for it in range(3):
            uv = torch.from_numpy(syn.make_pixels(128, g))[None].cuda()
            gt = {"rgb": torch.rand((128, 3), generator=g)[None].cuda(), "mask": torch.ones((1, 128, 3)).cuda()}
            bs.append(({"intrinsics": K, "uv": uv, "pose": torch.from_numpy(scene["poses"][it])[None].cuda(), "local_data": None}, gt))
        return model, bs

    class Seeded:
        def __init__(self, step, sd):
            self.step, self.sd, self.it = step, sd, 0

        def __call__(self, *b):
            torch.manual_seed(1000 * self.sd + self.it)
            self.it += 1
            return self.step(*b)

    seeds = (41, 42, 43, 44)
    alone = []
    for sd in seeds:
        model, bs = make(sd)
        step = Seeded(TrainStep(model, use_graph=True), sd)
        alone.append([float(step(*b)[0]["loss"].item()) for b in bs])
    built = [make(sd) for sd in seeds]
    multi = MultiSceneTrainer([Seeded(TrainStep(m, use_graph=True), sd) for (m, _), sd in zip(built, seeds)], n_streams=2, device="cuda")
    together = [[] for _ in seeds]
    for it in range(3):
        out = multi.step([bs[it] for _, bs in built])
        torch.cuda.synchronize()
        for s, l in enumerate(out):
            together[s].append(float(l["loss"].item()))       # read before the next replay overwrites the graph's static output
    for s in range(len(seeds)):
        np.testing.assert_allclose(together[s], alone[s], rtol=2e-5)


@pytest.mark.parametrize("n_points,spacing,n_rays", [(50000, 0.025, 1024), (200000, 0.0125, 4096)])
def test_large_configs_run_and_paths_agree(n_points, spacing, n_rays):
    """BASELINE configs 3 and 5 shapes (garden-like 5e4 points in the +-2 grid; dense 2e5-point cloud with 4096-ray batches):
    the step runs at full size, every output is finite and physically bounded, and the two independent execution paths
    (exact-size buffers + library GEMMs vs worst-case buffers + device-side counts) agree."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep

    scene = syn.make_scene(n_points, seed=21, spacing=spacing)
    assert tuple(scene["ranges"])[0] == -2.0
    scene["intrinsics"], scene["poses"] = syn.make_cameras(ring_radius=4.0)
    g = torch.Generator().manual_seed(6)
    uv = torch.from_numpy(syn.make_pixels(n_rays, g))[None].cuda()
    K, pose = torch.from_numpy(scene["intrinsics"])[None].cuda(), torch.from_numpy(scene["poses"][0])[None].cuda()
    gt = {"rgb": torch.rand((n_rays, 3), generator=g)[None].cuda(), "mask": torch.ones((1, n_rays, 3)).cuda()}
    res = []
    for sync_free in (False, True):
        model = build_model(scene)
        step = TrainStep(model, sync_free=sync_free, keep_grads=True)       # the test reads the gradient buffer after the step
        torch.manual_seed(2)
        losses, out = step({"intrinsics": K, "uv": uv, "pose": pose, "local_data": None}, gt)
        res.append((losses["loss"].item(), step.flat.buffer.clone(), out))
        w = out["weights"].detach()
        assert torch.isfinite(w).all() and float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
        assert torch.isfinite(out["rgb_values"]).all() and float(out["rgb_values"].max()) <= 1.0 + 1e-5
        assert torch.isfinite(step.flat.buffer).all()
        if not sync_free:
            assert model.stats["valid_points"] > 20 * n_rays, "the rays must actually hit the cloud"
    (l0, g0, _), (l1, g1, _) = res
    np.testing.assert_allclose(l1, l0, rtol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=5e-3, atol=5e-5 * float(g0.abs().max()))


def _zero_crossings(sdf, axes, b):
    """Points where the SDF changes sign along grid edges (linear interpolation), as marching cubes would place vertices."""
    pts = []
    n = len(axes)
    vol = sdf.reshape(n, n, n)
    valid = vol != 1000.0
    grid = np.stack(np.meshgrid(axes, axes, axes, indexing="ij"), -1)
    for ax in range(3):
        sl0 = [slice(None)] * 3
        sl1 = [slice(None)] * 3
        sl0[ax], sl1[ax] = slice(0, n - 1), slice(1, n)
        a, c = vol[tuple(sl0)], vol[tuple(sl1)]
        ok = valid[tuple(sl0)] & valid[tuple(sl1)] & (a * c < 0)
        t = (a / (a - c + 1e-30))[ok]
        p0, p1 = grid[tuple(sl0)][ok], grid[tuple(sl1)][ok]
        pts.append(p0 + t[:, None] * (p1 - p0))
    return np.concatenate(pts, 0)


def test_zero_level_set_chamfer_vs_oracle():
    """BASELINE's acceptance language is "equal Chamfer": the zero-level sets of the HIP SDF and of the oracle SDF on the same
    scene coincide (symmetric Chamfer distance ~ float round-off, far below the 0.025 point spacing)."""
    from scipy.spatial import cKDTree

    from oracle import path as P
    from spurfies_amd import synthetic as syn

    scene = syn.make_scene(6000, seed=0)
    # a prior whose SDF actually crosses zero: offset T's bias so that roughly half of the near-surface samples are negative
    model = build_model(scene, train=False)
    b = scene["base_radius"] * 1.25
    n = 40
    axes = np.linspace(-b, b, n).astype(np.float32)
    x = torch.from_numpy(np.stack(np.meshgrid(axes, axes, axes, indexing="ij"), -1).reshape(-1, 3))
    with torch.no_grad():
        s0 = model.get_sdf_eval(x.cuda()).cpu().numpy()
    med = float(np.median(s0[s0 != 1000.0]))
    st_np = dict(scene["state"])
    st_np["T.0.bias"] = st_np["T.0.bias"] - np.float32(med)
    model.load_state_dict({"T.0.bias": torch.from_numpy(st_np["T.0.bias"])}, strict=False)
    with torch.no_grad():
        s_hip = model.get_sdf_eval(x.cuda()).cpu().numpy()
    st = P.load_state(st_np, requires_grad=False)
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    with torch.no_grad():
        s_orc = P.sdf_at_points(x, P.make_grid(cfg, st["neural_pts"]), st, cfg)[0].numpy()
    assert np.array_equal(s_hip != 1000.0, s_orc != 1000.0)
    za, zb = _zero_crossings(s_hip, axes, b), _zero_crossings(s_orc, axes, b)
    assert len(za) > 500 and abs(len(za) - len(zb)) <= 2
    da, _ = cKDTree(zb).query(za)
    db, _ = cKDTree(za).query(zb)
    chamfer = 0.5 * (da.mean() + db.mean())
    assert chamfer < 1e-5, chamfer


def test_surface_route_grid_sweep_and_chamfer():
    """SURVEY §8(f) N3 front half: the reference-shaped evaluation grid (plots.py:302-333), the chunked get_sdf_eval sweep over
    ~3e5 points and the zero-level surface; the HIP surface coincides with the oracle's (Chamfer ~ round-off) and lies on the
    analytic shape the synthetic prior was built around only up to the prior's own error (sanity bound)."""
    from oracle import path as P
    from spurfies_amd import synthetic as syn
    from spurfies_amd.utils import surface

    scene = syn.make_scene(6000, seed=0)
    model = build_model(scene, train=False)
    b = scene["base_radius"] * 1.25
    grid = surface.get_grid(None, 64, input_min=np.array([-b, -b, -b]), input_max=np.array([b, b, b]), eps=0.0)
    assert grid["grid_points"].shape[0] >= 64 ** 3
    vol0 = surface.sdf_volume(model.get_sdf_eval, grid, splitn=100000)
    med = float(np.median(vol0[vol0 != 1000.0]))
    st_np = dict(scene["state"])
    st_np["T.0.bias"] = st_np["T.0.bias"] - np.float32(med)          # a prior whose level set crosses the cloud
    model.load_state_dict({"T.0.bias": torch.from_numpy(st_np["T.0.bias"])}, strict=False)
    vol = surface.sdf_volume(model.get_sdf_eval, grid, splitn=100000)
    # the model's own sweep (points generated on the device, occupancy pre-filter, 4 M-point chunks: PointVolSDF.sdf_eval_grid) against the
    # reference's formulation — the uploaded [M,3] array in 100 000-point chunks through get_sdf_eval: the same volume, bit for bit
    vol_ref_form = surface.sdf_volume(lambda p: model.get_sdf_eval(p), grid, splitn=100000)
    assert np.array_equal(vol, vol_ref_form)
    small = model.sdf_eval_grid(*grid["xyz"], chunk=5000, span=70001).cpu().numpy()      # several spans, ragged chunks
    assert np.array_equal(small, vol)
    pts = surface.surface_points(vol, grid)
    assert len(pts) > 2000
    st = P.load_state(st_np, requires_grad=False)
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    sub = grid["grid_points"][::7]
    with torch.no_grad():
        s_orc = P.sdf_at_points(sub, P.make_grid(cfg, st["neural_pts"]), st, cfg)[0].numpy()
    s_hip = vol.reshape(-1)[::7]
    ok = (s_orc != 1000.0)
    assert np.array_equal(ok, s_hip != 1000.0)
    np.testing.assert_allclose(s_hip[ok], s_orc[ok], rtol=1e-4, atol=5e-6)
    d, acc, comp = surface.chamfer(pts, pts[::-1].copy())
    assert d == 0.0
    # every surface point has neural points nearby (the SDF is only defined within the kNN radius of the cloud)
    from scipy.spatial import cKDTree
    dist, _ = cKDTree(scene["state"]["neural_pts"]).query(pts)
    assert float(dist.max()) <= 0.05 + 2 * (2 * b / 63)
    # own triangulation of the same volume: its vertices contain the edge crossings (all but those on the rim, whose cells touch a
    # no-neighbour sample and are skipped), no edge is shared by more than two faces
    verts, faces = surface.triangulate(vol, grid)
    dv, _ = cKDTree(verts).query(pts)
    assert len(faces) > 2 * len(pts) // 3 and float((dv < 1e-9).mean()) > 0.9
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0), 1)
    assert int(np.unique(e, axis=0, return_counts=True)[1].max()) <= 2


def test_chamfer_after_optimisation_steps_hip_vs_oracle_and_analytic_surface():
    """BASELINE's acceptance language is "equal Chamfer".  On the fitted-prior scene (the SDF is the signed distance to the analytic
    surface) after K = 3 optimisation steps on the GPU and on the oracle: the reference-style mesh route (grid sweep through
    get_sdf_eval, iso-surface, largest component, triangle sampling, greedy down-sampling, accuracy / completeness —
    plots.py:188-287, evals/eval_dtu.py:60-254) gives the same surface for both, and that surface is the analytic one."""
    from oracle import path as P
    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import TrainStep
    from spurfies_amd.utils import surface

    scene = syn.make_scene(6000, seed=3, prior="fitted")
    model = build_model(scene)
    step = TrainStep(model)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    ogrid = P.make_grid(cfg, st["neural_pts"])
    opt, sched = P.make_optimizer(st)
    g = torch.Generator().manual_seed(70)
    K = torch.from_numpy(scene["intrinsics"])[None]
    for it in range(3):
        uv = torch.from_numpy(syn.make_pixels(48, g))[None]
        pose = torch.from_numpy(scene["poses"][it % 3])[None]
        rgb, mask = torch.rand((48, 3), generator=g), torch.ones((48,))
        torch.manual_seed(200 + it)
        step({"intrinsics": K.cuda(), "uv": uv.cuda(), "pose": pose.cuda(), "local_data": None},
             {"rgb": rgb[None].cuda(), "mask": mask[None, :, None].repeat(1, 1, 3).cuda()})
        torch.manual_seed(200 + it)
        P.train_step_grads({"intrinsics": K, "uv": uv, "pose": pose}, rgb, mask, st, cfg, grid=ogrid)
        P.optimizer_step(st, opt, sched)
    model.eval()
    b = scene["base_radius"] * 1.25
    verts, faces, vol, grid = surface.extract_surface(model.get_sdf_eval, 48, [-b] * 3, [b] * 3)
    assert faces is not None and len(faces) > 5000

    def oracle_sdf(x):
        with torch.no_grad():
            return P.sdf_at_points(x.cpu(), ogrid, st, cfg)[0]

    overts, ofaces, ovol, _ = surface.extract_surface(oracle_sdf, 48, [-b] * 3, [b] * 3, device="cpu")
    assert np.array_equal(vol != 1000.0, ovol != 1000.0)
    h = 2 * b / 47
    pts_h, pts_o = surface.sample_mesh_points(verts, faces, 0.25 * h), surface.sample_mesh_points(overts, ofaces, 0.25 * h)
    same = surface.chamfer_dtu(pts_h, pts_o, max_dist=10 * h, thresh=0.25 * h, seed=0)
    assert same["overall"] < 0.3 * h, same            # both surfaces sampled at 0.25 h: coincident up to the sampling
    np.testing.assert_allclose(vol[vol != 1000.0], ovol[ovol != 1000.0], rtol=2e-3, atol=2e-5)       # after three Adam steps
    # against the analytic surface: exact points on the lobed sphere
    rng = np.random.default_rng(1)
    d = rng.standard_normal((40000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gt = d * syn.lobed_radius(d, scene["base_radius"])[:, None]
    res = surface.chamfer_dtu(pts_h, gt, max_dist=10 * h, thresh=0.25 * h, seed=0)
    assert res["accuracy"] < 0.01 and res["completeness"] < 0.01, res        # 0.025-spaced cloud, prior rmse 6e-4, grid step ~0.037


@pytest.mark.parametrize("graph", [False, True])
def test_full_image_stream_matches_the_reference_in_its_evaluation_configuration(graph):
    """eval_image_near0.npz (the imported reference: eval_spurfies.py:276-292 loop, sampler range of config/confs/dtu_pn.conf:48, near = 0.0,
    split_input chunks of 500 + a short last chunk, merge_output) against the streamed renderer (spurfies_amd/eval_graph.py:ImageRenderer:
    device-side cursor, outputs scattered into pre-allocated [H*W, ...] tensors, one hipGraph launch per chunk with graph=True).  Per pixel:
    most pixels at the single-step end-to-end bound, every pixel within 3e-2 (numbers below); the realised sampler iterations of every chunk
    are the reference's."""
    from spurfies_amd.eval_graph import ImageRenderer

    fx = load_golden("eval_image_near0.npz")
    scene = scene_of(fx)
    model = build_model(scene, train=False, near=float(fx["meta.near"]))
    inp = inputs_of(fx, scene, device="cuda")
    total, chunk = fx["in.uv"].shape[0], int(fx["meta.chunk"])
    r = ImageRenderer(model, chunk, fast=-1, graph=graph, keep_weights=True)
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    out = r(inp, total, iters=True)
    torch.cuda.synchronize()
    # sdf_importance calls of the reference's loop per chunk (ray_sampler.py:403) = the realised iterations as this build counts them
    assert r.last_iters == [int(c) for c in fx["meta.sampler_calls_per_chunk"]], (r.last_iters, fx["meta.sampler_calls_per_chunk"])
    # Measured on MI355X (tools/e2e_errors.py, round 4; eager and graphed identical): pixels (all channels / slots) within the single-step end-to-end
    # bound rtol 1e-4 / atol 2e-5: rgb 96.8 %, depth 99.3 %, normals 90.6 %, the 80 compositing weights 83.3 %; largest deviation of any pixel
    # 5.7e-3 (rgb), 9.8e-3 (depth), 1.1e-2 (normal), 1.25e-2 (one weight).  The full evaluation loop runs up to five sampler iterations of
    # exp / log / bisection per ray (the HIP sampler alone: 2e-4 per ray on >= 97 % of the rays against the reference's, test_gpu_stages),
    # so sample positions differ in more than the last bits on some rays; the 24-ray evaluation fixture (near 0.5) agrees to 5e-6.
    for k, tight_frac, loose in (("rgb_values", 0.94, 3e-2), ("depth_values", 0.97, 3e-2), ("weights", 0.75, 3e-2), ("normal_map", 0.85, 3e-2)):
        got, want = out[k].cpu().numpy().reshape(total, -1), fx[f"out.{k}"].reshape(total, -1)
        assert np.isfinite(got).all(), k
        ok = np.isclose(got, want, rtol=1e-4, atol=2e-5).all(axis=1)
        assert ok.mean() >= tight_frac, f"{k}: only {ok.mean():.3f} of the pixels within rtol 1e-4 / atol 2e-5"
        assert float(np.abs(got - want).max()) <= loose, f"{k}: a pixel is off by {float(np.abs(got - want).max()):.3e}"
    # a second image through the same renderer (cursor reset, buffers reused) is the same image up to the float atomics of the forward's
    # RBF-weighted mean (last-bit run-to-run noise in the default scatter mode)
    first = {k: v.clone() for k, v in out.items()}
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    again = r(inp, total)
    for k in first:
        close = torch.isclose(first[k], again[k], rtol=1e-3, atol=1e-4).all(dim=1)
        assert float(close.float().mean()) >= 0.995, k
