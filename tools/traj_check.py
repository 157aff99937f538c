"""Replay tests/golden/trajectory_ref.npz (200 reference optimisation steps) on the GPU and print the per-step deviations."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from tests.helpers import load_golden, scene_of, check_probes
from tests.test_gpu_model import build_model
from spurfies_amd.train import TrainStep

fx = load_golden("trajectory_ref.npz")
scene = scene_of(fx)
model = build_model(scene)
mode = sys.argv[1] if len(sys.argv) > 1 else "sync_free"
step = TrainStep(model, sync_free=(mode == "sync_free"))
K = torch.from_numpy(scene["intrinsics"])[None].cuda()
torch.manual_seed(int(fx["meta.seed"]) + 7)
n = int(fx["meta.steps"])
rows = []
before = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
for i in range(n):
    inp = {"intrinsics": K, "uv": torch.from_numpy(fx["step.uv"][i])[None].cuda(), "pose": torch.from_numpy(scene["poses"][int(fx["step.view"][i])])[None].cuda(),
           "local_data": None, "iter_step": i}
    gt = {"rgb": torch.from_numpy(fx["step.rgb_gt"][i])[None].cuda(), "mask": torch.from_numpy(fx["step.mask_gt"][i])[None, :, None].repeat(1, 1, 3).cuda()}
    losses, out = step(inp, gt)
    rows.append({k: float(v.item()) for k, v in losses.items()})
keys = [k[5:] for k in fx if k.startswith("loss.")]
for k in keys:
    ref = fx["loss." + k]
    got = np.asarray([r[k] for r in rows])
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
    print(k, "max rel dev first 10 / 50 / all: %.2e %.2e %.2e   final ref %.5f got %.5f" % (rel[:10].max(), rel[:50].max(), rel.max(), ref[-1], got[-1]))
for pname, p in model.named_parameters():
    if p.requires_grad:
        d = (p.detach() - before[pname]).cpu().double().reshape(-1)
        st = fx[f"delta.{pname}.stats"]
        idx = fx[f"delta.{pname}.idx"]
        val = fx[f"delta.{pname}.val"]
        err = np.abs(d.numpy()[idx] - val)
        print(pname, "norm ref %.4e got %.4e; probe err max %.2e, median %.2e, ref probe absmax %.2e" % (st[2], float(d.norm()), err.max(), np.median(err), np.abs(val).max()))
print("adam t", getattr(step.optimizer, "t", None), "skipped", getattr(step, "skipped", None))
