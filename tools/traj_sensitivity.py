"""How sensitive are the END-OF-RUN parameter-delta norms of the 200-step reference trajectory to last-bit perturbations?  CPU only, no HIP code
involved: the oracle (oracle/path.py, pinned to the reference step by step) replays tests/golden/trajectory_ref.npz
  (a) as it is, with the thread counts 1 and 8 (torch's CPU reductions change their summation order with the thread count), and
  (b) with every trainable parameter perturbed ONCE before step 0 by a relative 2^-22 (a quarter of an fp32 ulp-scale change of the values),
and prints |delta p| / |reference delta p| - 1 per trainable tensor.  If (a)/(b) land several per cent apart on the bias vectors, the 5 - 11 %
the HIP run shows there (tests/test_gpu_model.py) is the trajectory's own sensitivity, not a difference in the gradients.
    python3 tools/traj_sensitivity.py [steps] -> profiles/r04_traj_sensitivity.json"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import path as P  # noqa: E402
from tests.helpers import load_golden, scene_of  # noqa: E402

fx = load_golden("trajectory_ref.npz")
scene = scene_of(fx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else int(fx["meta.steps"])
K = torch.from_numpy(scene["intrinsics"])[None]


def run(threads, perturb):
    torch.set_num_threads(threads)
    st = P.load_state(scene["state"])
    cfg = P.PathConfig(ranges=tuple(scene["ranges"]))
    grid = P.make_grid(cfg, st["neural_pts"])
    trainable = {k: v for k, v in st.items() if getattr(v, "requires_grad", False)}
    if perturb:
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for v in trainable.values():
                v.mul_(1.0 + perturb * (2.0 * torch.rand(v.shape, generator=g) - 1.0))
    before = {k: v.detach().clone() for k, v in trainable.items()}
    opt, sched = P.make_optimizer(st)
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    losses = []
    for i in range(n):
        inp = {"intrinsics": K, "uv": torch.from_numpy(fx["step.uv"][i])[None], "pose": torch.from_numpy(scene["poses"][int(fx["step.view"][i])])[None], "local_data": None}
        _, l, _ = P.train_step_grads(inp, torch.from_numpy(fx["step.rgb_gt"][i]), torch.from_numpy(fx["step.mask_gt"][i]), st, cfg, grid=grid)
        P.optimizer_step(st, opt, sched)
        losses.append(float(l["loss"].item()))
    return np.asarray(losses), {k: (v.detach() - before[k]).double() for k, v in trainable.items()}


res = {"fixture": "trajectory_ref.npz", "steps": n, "runs": {}}
ref_loss = fx["loss.loss"][:n]
for tag, threads, perturb in (("oracle_1_thread", 1, 0.0), ("oracle_8_threads", 8, 0.0), ("oracle_perturbed_2^-22", 8, 2.0 ** -22)):
    t0 = time.time()
    losses, deltas = run(threads, perturb)
    rel = np.abs(losses - ref_loss) / np.abs(ref_loss)
    row = {"seconds": time.time() - t0, "loss_max_rel_first10": float(rel[:10].max()), "loss_max_rel_all": float(rel.max()), "tensors": {}}
    for k, d in deltas.items():
        key = f"delta.{k}.stats"
        if key in fx and n == int(fx["meta.steps"]):
            row["tensors"][k] = float(d.norm() / fx[key][2] - 1.0)
    res["runs"][tag] = row
    print(tag, f"{row['seconds']:.0f} s", "loss max rel dev first 10 / all: %.2e %.2e" % (row["loss_max_rel_first10"], row["loss_max_rel_all"]),
          " ".join(f"{k.split('.')[-2] if '.' in k else k}.{k.split('.')[-1]} {v * 100:+.2f}%" for k, v in row["tensors"].items() if "bias" in k), flush=True)
json.dump(res, open("profiles/r04_traj_sensitivity.json", "w"), indent=1)
