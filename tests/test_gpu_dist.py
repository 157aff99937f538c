"""GPU: the ray-sharded TrainStep (2 ranks, gloo, both on cuda:0 — the test box has one GPU) produces the same
summed gradients and loss as one process rendering all rays; and bench.py runs under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_TOTAL = 96


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_model():
    from spurfies_amd import synthetic as syn
    from spurfies_amd.conf import default_model_conf
    from spurfies_amd.model.pointneus_disent import PointVolSDF

    scene = syn.make_scene(3000, seed=4)
    st = scene["state"]
    conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
    g = torch.Generator().manual_seed(8)
    uv = torch.from_numpy(syn.make_pixels(R_TOTAL, g))
    rgb, mask = torch.rand((R_TOTAL, 3), generator=g), (torch.rand((R_TOTAL,), generator=g) > 0.2).float()
    K, pose = torch.from_numpy(scene["intrinsics"])[None].cuda(), torch.from_numpy(scene["poses"][1])[None].cuda()
    return model, uv, rgb, mask, K, pose


def _patch_draws(rank, world):
    """Every rank draws the batch-wide CPU random numbers and keeps its rays' rows, so that ranks and the
    single-process run see identical per-ray draws."""
    orig = torch.rand

    def rand(*shape, **kw):
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        if len(shp) == 2 and shp[0] == R_TOTAL // world and world > 1:
            out = kw.pop("out", None)                 # the sync-free step draws into pinned buffers
            mine = orig((R_TOTAL, shp[1]), **kw)[rank::world].contiguous()
            return mine if out is None else out.copy_(mine)
        return orig(*shape, **kw)

    torch.rand = rand


def _run_step(rank, world, sync_free=False):
    from spurfies_amd import dist as sdist
    from spurfies_amd.train import TrainStep

    model, uv, rgb, mask, K, pose = _setup_model()
    step = TrainStep(model, sync_free=sync_free)
    sel = sdist.shard_rays(R_TOTAL)
    torch.manual_seed(21)
    _patch_draws(rank, world)
    losses, _ = step({"intrinsics": K, "uv": uv[sel][None].cuda(), "pose": pose, "local_data": None},
                     {"rgb": rgb[sel][None].cuda(), "mask": mask[sel][None, :, None].repeat(1, 1, 3).cuda()})
    # TrainStep clipped + zero-guarded the flat buffer in place; recover the raw summed gradient direction
    total = losses["loss"].detach().clone()
    sdist.all_reduce_sum(total)
    return total.item(), step.flat.buffer.detach().cpu().numpy()


def _worker(rank, world, port, q, sync_free=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _run_step(rank, world, sync_free))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("sync_free", [False, True])
def test_two_ranks_on_one_gpu_match_single_process(sync_free):
    """sync_free=True exercises the fused loss kernels with all-reduced normalisers (spf_loss_forward's `denom`)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sync_free)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    loss1, g1 = _run_step(0, 1, sync_free)
    for _, loss2, g2 in res:
        np.testing.assert_allclose(loss2, loss1, rtol=2e-5)
        # clipped gradients (norm <= 1): identical up to float-atomic summation order
        np.testing.assert_allclose(g2, g1, rtol=5e-3, atol=2e-5 * float(np.abs(g1).max()))
    np.testing.assert_array_equal(res[0][2], res[1][2])   # replicas stay bit-identical after the all-reduce


def test_bench_under_torchrun_two_ranks_gloo():
    env = dict(os.environ, SPF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rays", "128", "--points", "3000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["config"]["rays_per_gpu"] == 128
