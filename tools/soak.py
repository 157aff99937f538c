"""Soak run: N sync-free optimisation steps on the bench scene, cycling batches; reports loss trajectory, skipped (non-finite) updates and timing."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline", "--sustained", "0"]
import bench  # noqa: E402

n_steps = 20000
args = bench.parse()
dev = torch.device("cuda", 0)
torch.set_num_threads(1)
scene, model, step = bench.build_scene_step(args, 0, dev, 1, False)
batches = bench.make_batches(scene, 64, 1024, 0, 1, dev)
torch.manual_seed(1)
hist = []
t0 = time.perf_counter()
for i in range(n_steps):
    losses, _ = step(*batches[i % 64])
    if i % 2000 == 0 or i == n_steps - 1:
        hist.append((i, float(losses["loss"].item()), float(losses["rgb_loss"].item())))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
state = step.optimizer._flat["state"].tolist()
print("steps", n_steps, "ms/step %.3f" % (dt / n_steps * 1e3), "adam t", state[0], "skipped updates", state[1])
for h in hist:
    print("  step %6d loss %.5f rgb %.5f" % h)
ok = all(np.isfinite(h[1]) for h in hist) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
print("finite:", ok, "beta", float(model.density.get_beta()))
