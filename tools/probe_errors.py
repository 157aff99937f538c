"""How much of the end-to-end gradient-probe tolerance (tests/test_gpu_model.py: rtol 2e-2, atol 2e-2 x the tensor's RMS gradient) do the HIP
gradients actually use, per trainable tensor — with float atomics and with the bit-reproducible scatter mode?  For the two single-step fixtures
with the sampler in the loop.   python3 tools/probe_errors.py -> gpurun_out/probe_errors.json"""
import json
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spurfies_amd import ops  # noqa: E402
from spurfies_amd.model.loss import VolSDFLoss  # noqa: E402
from spurfies_amd.train import TrainStep  # noqa: E402
from tests.helpers import inputs_of, load_golden, scene_of  # noqa: E402
from tests.test_gpu_model import build_model  # noqa: E402

res = {}
for name in ("step_train_r128.npz", "step_train_far.npz"):
    fx = load_golden(name)
    scene = scene_of(fx)
    for mode, sync_free in (("atomic", False), ("atomic", True), ("fixed", True)):
        ops.set_scatter_mode(mode)
        try:
            model = build_model(scene)
            inp = inputs_of(fx, scene, device="cuda")
            gt = {"rgb": torch.from_numpy(fx["in.rgb_gt"])[None].cuda(), "mask": torch.from_numpy(fx["in.mask_gt"])[None, :, None].repeat(1, 1, 3).cuda()}
            torch.manual_seed(int(fx["meta.seed"]) + 7)
            if sync_free:
                step = TrainStep(model, sync_free=True, keep_grads=True)
                step._forward_backward(inp, gt)
            else:
                out = model(inp, fast=1)
                loss_fn = VolSDFLoss("torch.nn.L1Loss", local_weight=0.5, pseudo_weight=0.5, eikonal_weight=0.001, rgb_weight=1.0, tv_weight=0.01)
                model.zero_grad()
                loss_fn(out, {k: v.cpu() for k, v in gt.items()})["loss"].backward()
            row = {}
            for pname, p in model.named_parameters():
                if not p.requires_grad:
                    continue
                g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).double().cpu().numpy()
                idx, val, st = fx[f"grad.{pname}.idx"], fx[f"grad.{pname}.val"].astype(np.float64), fx[f"grad.{pname}.stats"]
                scale = float(st[2]) / max(np.sqrt(g.size), 1.0)
                err = np.abs(g[idx] - val)
                # the smallest rtol r such that err <= r |ref| + r scale holds on every probe (the test's form with rtol = atol factor = r)
                need = float((err / (np.abs(val) + scale + 1e-30)).max())
                row[pname] = {"rtol_needed": need, "l2_rel": float(abs(np.linalg.norm(g) - st[2]) / max(st[2], 1e-30))}
            res[f"{name}:{mode}:{'sync_free' if sync_free else 'default'}"] = row
            worst = max(row.items(), key=lambda kv: kv[1]["rtol_needed"])
            print(name, mode, "sync_free" if sync_free else "default", "worst tensor", worst[0], "rtol needed %.2e" % worst[1]["rtol_needed"],
                  "| max l2 rel %.2e" % max(v["l2_rel"] for v in row.values()), flush=True)
        finally:
            ops.set_scatter_mode("atomic")
json.dump(res, open("gpurun_out/probe_errors.json", "w"), indent=1)
