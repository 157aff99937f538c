"""Round-3 verdict item 8: is the 5 - 11 % drift of the bias vectors' total change over the 200-step reference trajectory float-atomic noise, or a real
difference?  Replays tests/golden/trajectory_ref.npz twice with set_scatter_mode("fixed") (bit-reproducible steps) and twice with the default float
atomics, and prints, per trainable tensor, |delta| / |reference delta| - 1 of each run, whether the two fixed runs agree bit for bit, and the
probe-wise error of the parameter deltas.   python3 tools/traj_fixed.py [fixture.npz] -> gpurun_out/traj_fixed.json"""
import json
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spurfies_amd import ops  # noqa: E402
from spurfies_amd.train import TrainStep  # noqa: E402
from tests.helpers import load_golden, scene_of  # noqa: E402
from tests.test_gpu_model import build_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "trajectory_ref.npz"
fx = load_golden(name)
scene = scene_of(fx)
K = torch.from_numpy(scene["intrinsics"])[None].cuda()
n = int(fx["meta.steps"])


def run(mode):
    ops.set_scatter_mode(mode)
    try:
        model = build_model(scene)
        step = TrainStep(model, sync_free=True)
        before = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
        torch.manual_seed(int(fx["meta.seed"]) + 7)
        losses = []
        for i in range(n):
            inp = {"intrinsics": K, "uv": torch.from_numpy(fx["step.uv"][i])[None].cuda(), "pose": torch.from_numpy(scene["poses"][int(fx["step.view"][i])])[None].cuda(),
                   "local_data": None, "iter_step": i}
            gt = {"rgb": torch.from_numpy(fx["step.rgb_gt"][i])[None].cuda(), "mask": torch.from_numpy(fx["step.mask_gt"][i])[None, :, None].repeat(1, 1, 3).cuda()}
            l, _ = step(inp, gt)
            losses.append(l["loss"])
        losses = np.asarray([float(v.item()) for v in losses])
        deltas = {k: (p.detach() - before[k]).double().cpu() for k, p in model.named_parameters() if p.requires_grad}
        return losses, deltas
    finally:
        ops.set_scatter_mode("atomic")


runs = {"fixed_1": run("fixed"), "fixed_2": run("fixed"), "atomic_1": run("atomic"), "atomic_2": run("atomic")}
res = {"fixture": name, "steps": n, "tensors": {}, "loss": {}}
ref_loss = fx["loss.loss"]
for tag, (losses, _) in runs.items():
    rel = np.abs(losses - ref_loss) / np.abs(ref_loss)
    res["loss"][tag] = {"max_rel_first10": float(rel[:10].max()), "max_rel_first50": float(rel[:50].max()), "max_rel_all": float(rel.max()), "final": float(losses[-1])}
res["loss"]["reference_final"] = float(ref_loss[-1])
same = all(torch.equal(runs["fixed_1"][1][k], runs["fixed_2"][1][k]) for k in runs["fixed_1"][1]) and np.array_equal(runs["fixed_1"][0], runs["fixed_2"][0])
res["fixed_runs_bit_identical"] = bool(same)
for k in runs["fixed_1"][1]:
    st, idx, val = fx[f"delta.{k}.stats"], fx[f"delta.{k}.idx"], fx[f"delta.{k}.val"]
    row = {"reference_norm": float(st[2])}
    for tag, (_, deltas) in runs.items():
        d = deltas[k].reshape(-1)
        err = np.abs(d.numpy()[idx] - val)
        row[tag] = {"norm_ratio_minus_1": float(d.norm() / st[2] - 1.0), "probe_err_max_over_absmax": float(err.max() / max(np.abs(val).max(), 1e-30)),
                    "probe_err_median_over_absmax": float(np.median(err) / max(np.abs(val).max(), 1e-30))}
    res["tensors"][k] = row
    print(k, "  ".join(f"{tag} {row[tag]['norm_ratio_minus_1'] * 100:+.2f}%" for tag in runs), flush=True)
print("fixed runs bit-identical:", same)
print(json.dumps(res["loss"]))
json.dump(res, open("gpurun_out/traj_fixed.json", "w"), indent=1)
