"""Which kernel's arithmetic makes a late-training step's gradient non-finite?  Runs the soak's step (graph mode, local term) until updates start being
skipped, then replays the NEXT batches with the parameters frozen under each arithmetic configuration and lists the parameters whose gradient is not finite.
    python3 tools/h2_overflow_probe.py [--max-steps 12000] [--out gpurun_out/h2_overflow_probe.json]"""
import argparse
import json
import sys

import torch

sys.path.insert(0, ".")
_own = argparse.ArgumentParser()
_own.add_argument("--max-steps", type=int, default=12000)
_own.add_argument("--want-skipped", type=float, default=30)
_own.add_argument("--out", default="gpurun_out/h2_overflow_probe.json")
own, rest = _own.parse_known_args()
sys.argv = ["bench.py", "--no-cpu-baseline", "--sustained", "0"] + rest
import bench  # noqa: E402
from spurfies_amd import ops  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
torch.set_num_threads(1)
scene, model, step = bench.build_scene_step(args, 0, dev, 1, True)
batches = bench.make_batches(scene, 64, args.rays, 0, 1, dev, local=True)
torch.manual_seed(1)
i = 0
while i < own.max_steps:
    for _ in range(250):
        step(*batches[i % 64])
        i += 1
    st = step.optimizer._flat["state"].tolist()
    print("step", i, "adam_t", st[0], "skipped", st[1], flush=True)
    if st[1] >= own.want_skipped:
        break
blob = step.state_dict()
del step
from spurfies_amd.train import TrainStep  # noqa: E402

probe = TrainStep(model, sync_free=True, use_graph=False, keep_grads=True)
probe.load_state_dict(blob)
res = {"snapshot_step": i, "configs": {}}
configs = {"default": {}, "geo_split_w": {"geo": "split_w"}, "color_fwd_bf16x3": {"h2": {"color_fwd": False}}, "color_bwd_bf16x3": {"h2": {"color_bwd": False}},
           "wgrad_bf16x3": {"h2": {"wgrad": False}}, "all_bf16x3": {"geo": "split_w", "h2": {"color_fwd": False, "color_bwd": False, "wgrad": False}}}
for name, cfg in configs.items():
    prev_geo = ops.geo_mode()
    ops.set_geo_mode(cfg.get("geo", prev_geo))
    prev = ops.set_h2(**cfg.get("h2", {}))
    bad_steps, bad_params, worst = 0, {}, {}
    for b in range(64):
        rng = torch.get_rng_state()
        torch.manual_seed(1000 + b)
        losses, out = probe._forward_backward(*batches[(i + b) % 64])
        torch.set_rng_state(rng)
        nonfinite = False
        for pn, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad
            fin = torch.isfinite(g)
            if not bool(fin.all()):
                nonfinite = True
                bad_params[pn] = bad_params.get(pn, 0) + 1
            m = float(g[fin].abs().max()) if bool(fin.any()) else 0.0
            worst[pn] = max(worst.get(pn, 0.0), m)
        bad_steps += int(nonfinite)
        if not bool(torch.isfinite(losses["loss"])):
            bad_params["<loss>"] = bad_params.get("<loss>", 0) + 1
    res["configs"][name] = {"steps_with_nonfinite_gradient_of_64": bad_steps, "parameters": bad_params,
                            "largest_finite_gradient_entries": dict(sorted(worst.items(), key=lambda kv: -kv[1])[:6])}
    print(name, res["configs"][name], flush=True)
    ops.set_h2(**prev)
    ops.set_geo_mode(prev_geo)
# parameter magnitudes at the snapshot
res["param_abs_max"] = {pn: float(p.detach().abs().max()) for pn, p in model.named_parameters()}
print({k: round(v, 3) for k, v in res["param_abs_max"].items()})
json.dump(res, open(own.out, "w"), indent=1)
