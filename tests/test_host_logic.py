"""CPU: host-side logic of the path that needs no kernel — CPU-generator draw sharding, SDF zero crossings, the local
(feature-consistency) loss against the oracle, the Chamfer / component utilities."""
import numpy as np
import torch

from oracle import path as P
from tests.helpers import load_golden, local_data_of, scene_of


def test_sharded_ranks_consume_the_single_gpu_draws():
    """every rank issues the reference's torch.rand calls for the whole batch and keeps rows rank::world, so
    the ranks' draws interleave to exactly what one GPU would have drawn, and the generator ends in the same state."""
    from spurfies_amd.model.ray_sampler import ErrorBoundSampler_pn

    def sampler(shard):
        s = ErrorBoundSampler_pn(3.0, near=0.5, far=4.5, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                                 max_total_iters=5)
        s.shard = shard
        return s

    torch.manual_seed(7)
    full = [sampler(None)._rand_rows(96, 128), sampler(None)._rand_rows(96, 64)]
    end_state = torch.rand(1)
    for world in (2, 4, 8):
        parts = []
        for rank in range(world):
            torch.manual_seed(7)
            s = sampler((rank, world))
            parts.append([s._rand_rows(96 // world, 128), s._rand_rows(96 // world, 64)])
            assert torch.equal(torch.rand(1), end_state)
        for q in range(2):
            merged = torch.empty_like(full[q])
            for rank in range(world):
                merged[rank::world] = parts[rank][q]
            assert torch.equal(merged, full[q])


def test_find_surface_points_and_local_loss_match_oracle_on_reference_stage_tensors():
    """PointVolSDF.find_surface_points (dense rows, branch-free) and feat_utils (sync-free masked form + the reference-signature
    get_local_loss) on the REFERENCE's recorded SDF / depth rows: crossings, depths and the loss value."""
    from spurfies_amd import feat_utils
    from spurfies_amd.model.pointneus_disent import PointVolSDF

    fx = load_golden("step_train_local.npz")
    scene = scene_of(fx)
    mask, ray_mask = fx["stage.mask"], fx["stage.ray_mask"]
    R, SR = mask.shape
    sdf = np.full((R, SR), 1000.0, np.float32)
    sdf[mask] = fx["stage.agg_sdf"][:, 0]
    z = np.zeros((R, SR), np.float32)
    z[ray_mask] = fx["stage.z_slots"][..., 0]
    d, hit = PointVolSDF.find_surface_points(torch.from_numpy(sdf), torch.from_numpy(z))
    assert np.array_equal(hit.numpy()[ray_mask], fx["stage.network_mask"][0]) and not hit.numpy()[~ray_mask].any()
    np.testing.assert_allclose(d.numpy()[ray_mask], fx["stage.d_surface"][0], rtol=1e-6, atol=1e-7)
    od, ohit = P.find_surface_points(torch.from_numpy(sdf), torch.from_numpy(z))
    assert torch.equal(ohit, hit)
    np.testing.assert_allclose(d.numpy(), od.numpy(), rtol=1e-6, atol=1e-7)
    # the loss on those surface points: masked dense form == reference-signature compacted form == recorded value
    local = local_data_of(fx, scene)
    o, dirs = torch.from_numpy(fx["stage.cam_loc"]), torch.from_numpy(fx["stage.ray_dirs"])
    pts = o + dirs * d[:, None]
    s, c = feat_utils.local_loss_terms(pts, hit, local)
    np.testing.assert_allclose(float(s / c), float(fx["out.local_loss"]), rtol=2e-5)
    ref_form = feat_utils.get_local_loss(pts[hit], None, local["feat"][None], local["cam"][None], local["feat_src"][None], local["src_cams"][None],
                                         local["size"].reshape(1), local["center"].reshape(1, 3), hit, hit)
    np.testing.assert_allclose(float(ref_form), float(fx["out.local_loss"]), rtol=2e-5)
    # no crossing anywhere -> exactly 0, and no NaN reaches the gradient of rays without a crossing
    sdf_t = torch.from_numpy(np.abs(sdf)).requires_grad_(True)
    d0, hit0 = PointVolSDF.find_surface_points(sdf_t, torch.from_numpy(z))
    assert not hit0.any() and float(d0.abs().sum()) == 0.0
    d0.sum().backward()
    assert torch.isfinite(sdf_t.grad).all()


def test_knn_compat_rules_of_the_oracle():
    """oracle/voxel_grid.py compat switches: the capacity limits keep the lowest-index points / cells, the layered search returns
    own-cell neighbours only when the own cell has k of them — checked against first principles on a dense cloud."""
    from oracle.voxel_grid import VoxelGridOracle
    from spurfies_amd import synthetic as syn

    pts, _, _ = syn.make_cloud(6000, spacing=0.012, seed=3)
    args = ((0.025,) * 3, (3,) * 3, (3,) * 3, 5, 40, (-1, -1, -1, 1, 1, 1))
    exact = VoxelGridOracle(*args)
    exact.set_pointset(pts)
    cells = exact.cell_of(pts)
    lin = (cells[:, 0] * exact.dims[1] + cells[:, 1]) * exact.dims[2] + cells[:, 2]
    tr = VoxelGridOracle(*args, compat=("truncate",))
    tr.set_pointset(pts)
    assert np.array_equal(tr.dims, exact.dims) and np.array_equal(tr.origin, exact.origin)
    kept = tr.in_range
    first_idx = {}
    for i, c in enumerate(lin):
        first_idx.setdefault(int(c), i)
    allowed = set(sorted(first_idx, key=first_idx.get)[:40])               # the 40 cells claimed first in index order
    for c in np.unique(lin):
        members = np.nonzero(lin == c)[0]
        want = members[:5] if int(c) in allowed else members[:0]           # their 5 lowest-index points
        assert np.array_equal(np.nonzero(kept & (lin == c))[0], want)
    x = (pts[::7] + np.float32(0.004)).astype(np.float32)
    lay = VoxelGridOracle(*args, compat=("layered",))
    lay.set_pointset(pts)
    a, b = exact.knn(x, 8, exact.radius(2)), lay.knn(x, 8, lay.radius(2))
    xc = exact.cell_of(x)
    n_own_only = 0
    for m in range(len(x)):
        d2 = ((pts - x[m]) ** 2).sum(1)
        own = np.all(cells == xc[m], axis=1) & (d2 <= float(exact.radius(2)) ** 2 * (1 + 1e-6))
        if own.sum() >= 8 + 2:                                             # clearly >= k in the own cell (margin for the fp32 radius test)
            assert set(b[m][b[m] >= 0]) <= set(np.nonzero(np.all(cells == xc[m], axis=1))[0])
            n_own_only += 1
        elif own.sum() < 8 - 2:
            assert np.array_equal(a[m], b[m])
    assert n_own_only > 20 and not np.array_equal(a, b)


def test_mesh_route_back_half_on_an_analytic_sphere():
    """SURVEY.md section 8(f) N3, back half (evals/eval_dtu.py:60-254, plots.py:213-215) on a case with a known answer: two spheres,
    the small one is discarded by the largest-component rule; triangle sampling covers the surface at the requested density; greedy
    down-sampling leaves no two points closer than the threshold; accuracy / completeness against exact sphere points are at the
    discretisation error and zero for identical clouds."""
    from spurfies_amd.utils import surface

    n = 40
    ax = np.linspace(-1.0, 1.0, n)
    grid = {"xyz": [ax, ax, ax]}
    gx, gy, gz = np.meshgrid(ax, ax, ax)
    big = np.sqrt(gx ** 2 + gy ** 2 + gz ** 2) - 0.5
    small = np.sqrt((gx - 0.8) ** 2 + (gy - 0.8) ** 2 + (gz - 0.8) ** 2) - 0.12
    vol = np.minimum(big, small).astype(np.float32)
    verts, faces = surface.triangulate(vol, grid)
    v1, f1 = surface.largest_component(verts, faces)
    assert len(f1) < len(faces) and np.abs(np.linalg.norm(v1, axis=1) - 0.5).max() < 0.01      # only the big sphere is left
    e = np.sort(np.concatenate([f1[:, [0, 1]], f1[:, [1, 2]], f1[:, [2, 0]]], 0), 1)
    assert (np.unique(e, axis=0, return_counts=True)[1] == 2).all()                             # and it is closed
    pts = surface.sample_mesh_points(v1, f1, 0.01)
    assert len(pts) > 3 * len(v1) and np.abs(np.linalg.norm(pts, axis=1) - 0.5).max() < 0.01
    down = surface.downsample_points(pts, 0.03, seed=1)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(down).query(down, k=2)
    assert d[:, 1].min() >= 0.03 and len(down) < len(pts) / 3
    assert cKDTree(down).query(pts)[0].max() <= 0.03 + 1e-12                                   # nothing struck out without a keeper nearby
    rng = np.random.default_rng(0)
    gt = rng.standard_normal((20000, 3))
    gt = 0.5 * gt / np.linalg.norm(gt, axis=1, keepdims=True)
    res = surface.chamfer_dtu(pts, gt, max_dist=0.2, thresh=0.02, seed=0)
    assert res["accuracy"] < 0.01 and res["completeness"] < 0.015 and abs(res["overall"] - 0.5 * (res["accuracy"] + res["completeness"])) < 1e-15
    same = surface.chamfer_dtu(gt, gt)
    assert same["accuracy"] == 0.0 and same["completeness"] == 0.0
    far = surface.chamfer_dtu(np.concatenate([gt, [[5.0, 5.0, 5.0]]]), gt, max_dist=0.2)         # outliers beyond max_dist are not averaged in
    assert far["accuracy"] == 0.0


def test_workspace_owner_tokens_are_never_reused_and_die_with_their_owner():
    """Scratch buffers are keyed by a per-owner token (ops.scratch_owner), not by id(owner): Python hands a dead object's id to the next
    allocation, and a new optimisation step must not inherit (or keep alive) a dead one's workspaces; the caches are dropped with the owner."""
    import gc

    from spurfies_amd import ops

    class Owner:
        pass

    a = Owner()
    with ops.scratch_owner(a):
        ta = ops._SCRATCH_OWNER[0]
        with ops.scratch_owner(Owner()):                       # nested owners restore the outer one
            assert ops._SCRATCH_OWNER[0] != ta
        assert ops._SCRATCH_OWNER[0] == ta
    assert ops._SCRATCH_OWNER[0] is None
    ops._wgrad_ws[(0, ("owner", ta, None), 123)] = "workspace of a"
    ops._fixed_bufs[(0, ("owner", ta, "geo_scatter"), "geo_latents")] = "accumulators of a"
    with ops.scratch_owner(a):
        assert ops._SCRATCH_OWNER[0] == ta                     # the same owner keeps its token
    del a
    gc.collect()
    assert not any(k[1][:2] == ("owner", ta) for k in list(ops._wgrad_ws) + list(ops._fixed_bufs) if isinstance(k[1], tuple))
    b = Owner()
    with ops.scratch_owner(b):
        assert ops._SCRATCH_OWNER[0] > ta                      # never handed out again


def test_capture_guard_keeps_the_collector_out_and_restores_it():
    """ops.capture_guard: collect before a hipGraph capture, cyclic collector off during it (a dead model's VoxelGrid.__del__ -> hipFree inside a
    capture invalidates it), previous state restored afterwards — also when the body raises."""
    import gc

    from spurfies_amd import ops

    assert gc.isenabled()
    with ops.capture_guard():
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with ops.capture_guard():
            raise RuntimeError("capture failed")
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with ops.capture_guard():
            assert not gc.isenabled()
        assert not gc.isenabled()                              # it was off before: stays off
    finally:
        gc.enable()


def test_deferred_weight_gradient_problems_do_not_outlive_an_abandoned_backward():
    """The head stage's weight-gradient problems wait in ops._PENDING_WGRAD for the colour trunk's launch; if a backward is abandoned between the
    two (an exception), the next forward drops them instead of launching stale pointers with the next step."""
    from spurfies_amd import ops

    ops._PENDING_WGRAD.append((["stale problem"], ("tensors",)))
    ops.drop_pending_wgrad()
    assert ops._PENDING_WGRAD == []
