"""CPU, world_size 2 over gloo: ray sharding + count-normalised losses + one flat-buffer all-reduce
reproduce the single-process batch loss and gradients exactly."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spurfies_amd import dist as sdist
from spurfies_amd.model.loss import VolSDFLoss


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss_mod():
    return VolSDFLoss("torch.nn.L1Loss", local_weight=0.5, pseudo_weight=0.5, eikonal_weight=0.001, rgb_weight=1.0, tv_weight=0.01)


def _toy_params():
    g = torch.Generator().manual_seed(0)
    return [torch.randn((6, 3), generator=g).requires_grad_(True), torch.randn((6, 5), generator=g).requires_grad_(True),
            torch.randn((4,), generator=g).requires_grad_(True)]


def _toy_forward(params, feats, n_pts_per_ray):
    """A differentiable stand-in for PointVolSDF.forward with the same output keys and data-dependent counts."""
    a, b, c = params
    rgb = torch.sigmoid(feats @ a)
    weights = torch.softmax(feats @ b, -1) * 0.9
    g = (feats.repeat_interleave(n_pts_per_ray, 0)[:, :3] * c[:3]) + c[3]
    valid = feats[:, 0] > 0
    sd = (feats @ a).sum(-1)
    return {"rgb_values": rgb, "weights": weights, "grad_theta": g, "tv_loss": (c ** 2).sum(), "local_loss": torch.tensor(0.0),
            "pseudo_sum": torch.where(valid, sd.abs(), torch.zeros_like(sd)).sum(), "pseudo_count": valid.sum(),
            "pseudo_pts_loss": torch.where(valid, sd.abs(), torch.zeros_like(sd)).sum() / valid.sum().clamp(min=1)}


def _data(R=64):
    g = torch.Generator().manual_seed(1)
    feats = torch.randn((R, 6), generator=g)
    n_pts = torch.randint(1, 5, (R,), generator=g)
    rgb_gt = torch.rand((R, 3), generator=g)
    mask = (torch.rand((R,), generator=g) > 0.3).float()
    return feats, n_pts, rgb_gt, mask


def _worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _toy_params()
        flat = sdist.FlatGrads(params)
        feats, n_pts, rgb_gt, mask = _data()
        sel = sdist.shard_rays(feats.shape[0])
        assert sel.tolist() == list(range(rank, feats.shape[0], world))
        out = _toy_forward(params, feats[sel], n_pts[sel])
        gt = {"rgb": rgb_gt[sel][None], "mask": mask[sel][None, :, None].repeat(1, 1, 3)}
        losses = sdist.sharded_loss(_loss_mod(), out, gt)
        flat.zero_()
        losses["loss"].backward()
        total = losses["loss"].detach().clone()
        sdist.all_reduce_sum(total)
        sdist.all_reduce_sum(flat.buffer)
        out_q.put((rank, total.item(), flat.buffer.clone().numpy(), [p.grad.data_ptr() == flat.buffer.data_ptr() + 4 * o for p, o in
                                                                      zip(params, np.cumsum([0] + [p.numel() for p in params[:-1]]))]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_process_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch, the reference's own loss normalisation
    params = _toy_params()
    feats, n_pts, rgb_gt, mask = _data()
    out = _toy_forward(params, feats, n_pts)
    losses = _loss_mod()(out, {"rgb": rgb_gt[None], "mask": mask[None, :, None].repeat(1, 1, 3)})
    losses["loss"].backward()
    want = np.concatenate([p.grad.reshape(-1).numpy() for p in params])
    for rank, total, buf, views_ok in res:
        assert all(views_ok), "every parameter's .grad must be a view into the flat buffer"
        np.testing.assert_allclose(total, losses["loss"].item(), rtol=1e-6)
        np.testing.assert_allclose(buf, want, rtol=2e-5, atol=1e-7)


def test_single_process_helpers():
    assert sdist.world_size() == 1 and sdist.rank() == 0
    assert sdist.shard_rays(5).tolist() == [0, 1, 2, 3, 4]
    t = torch.ones(3)
    assert sdist.all_reduce_sum(t) is t


def test_multi_scene_trainer_round_robin_cpu():
    """BASELINE.json configs[3] driver logic without a GPU: every scene's step callable sees exactly its own batch, once per round,
    in scene order; a batch-count mismatch is refused."""
    import pytest

    from spurfies_amd.train import MultiSceneTrainer

    seen = []

    def make(s):
        def step(inp, gt):
            seen.append((s, inp, gt))
            return {"loss": torch.tensor(float(s))}, None
        return step

    multi = MultiSceneTrainer([make(s) for s in range(11)], n_streams=2, device=None)
    assert len(multi) == 11
    for rnd in range(2):
        out = multi.step([(f"in{s}.{rnd}", f"gt{s}.{rnd}") for s in range(11)])
        assert [float(o["loss"]) for o in out] == [float(s) for s in range(11)]
    assert seen == [(s, f"in{s}.{r}", f"gt{s}.{r}") for r in range(2) for s in range(11)]
    with pytest.raises(ValueError):
        multi.step([("a", "b")])


PARAM_NAMES = ["neural_feats_color", "neural_feats_geometry", "F_color.0.weight", "F_color.0.bias", "F_color.2.weight", "F_color.2.bias",
               "F_color.4.weight", "F_color.4.bias", "F_color.6.weight", "F_color.6.bias", "R.0.weight", "R.0.bias", "R.2.weight", "R.2.bias",
               "R.4.weight", "R.4.bias", "density.beta"]


def _bucket_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        params = [torch.zeros(int(n), requires_grad=True) for n in (640, 320, 96, 8, 64, 8, 64, 8, 64, 8, 72, 8, 64, 8, 24, 3, 1)]
        flat = sdist.FlatGrads(params)
        flat.buffer.copy_(torch.randn(flat.buffer.shape, generator=g))
        mine = flat.buffer.clone()
        buckets = sdist.BucketedAllReduce(flat, PARAM_NAMES)
        buckets.begin()
        for name in ("head", "color_latents", "color_weights"):          # the order the backward announces them; geo_latents is left to finish()
            buckets.ready(name)
        buckets.ready("head")                                            # announcing a bucket twice must not reduce it twice
        buckets.finish()
        out_q.put((rank, mine.numpy(), flat.buffer.clone().numpy(), list(buckets.log), {k: [tuple(r) for r in v] for k, v in buckets.ranges.items()}))
    finally:
        dist.destroy_process_group()


def test_bucketed_all_reduce_equals_one_flat_all_reduce():
    """spurfies_amd/dist.py:BucketedAllReduce — the four gradient buckets, reduced one by one in the order the backward completes them,
    give exactly the sum one flat all-reduce gives; every element belongs to exactly one bucket."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for _, _, summed, log, ranges in res:
        np.testing.assert_array_equal(summed, total)
        assert log == ["head", "color_latents", "color_weights", "geo_latents"]
        cover = np.zeros(len(total), np.int32)
        for rs in ranges.values():
            for lo, hi in rs:
                cover[lo:hi] += 1
        assert (cover == 1).all()
        assert ranges["color_latents"] == [(0, 640)] and ranges["geo_latents"] == [(640, 960)] and len(ranges["color_weights"]) == 1 and len(ranges["head"]) == 1


def test_unknown_parameters_are_only_reduced_by_finish():
    """Round-3 advisor finding: a trainable tensor outside the head whitelist (an unfrozen prior layer, learnable points, a new module) must not
    ride in 'head' — the bucket reduced earliest — but in one only finish() reduces, after the whole backward."""
    B = sdist.BucketedAllReduce.bucket_of
    assert [B(n) for n in PARAM_NAMES] == (["color_latents", "geo_latents"] + ["color_weights"] * 6 + ["head"] * 9)
    for n in ("F_geometry.0.weight", "T.0.bias", "neural_pts", "extra.module.weight"):
        assert B(n) == "rest"
    params = [torch.zeros(4, requires_grad=True) for _ in range(3)]
    flat = sdist.FlatGrads(params)
    b = sdist.BucketedAllReduce(flat, ["R.0.weight", "F_geometry.0.weight", "density.beta"])
    assert b.ranges == {"head": [[0, 4], [8, 12]], "rest": [[4, 8]]} and b.bytes_per_step() == {"head": 32, "rest": 16}
    b.begin()
    assert b.armed
    b.finish()                       # world 1: nothing to reduce, but the step is disarmed
    assert not b.armed


def test_bench_without_launcher_refuses_rccl_ranks_it_has_no_gpus_for():
    """bench.py --gpus N with no WORLD_SIZE starts its own ranks; over RCCL it first checks that N GPUs are visible (none in this container)
    and says so instead of dying inside the first collective."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPF_DIST_BACKEND")}
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "visible GPUs" in out.stderr and not out.stdout.strip()
