"""Where the HOST's time goes in one graph-replayed optimisation step (128 rays per GPU: the GPU needs ~0.9 ms, the host must stay under that).
cProfile over 300 steps, the heaviest callees by cumulative time.  Usage (GPU box): python3 tools/host_step_profile.py [rays]"""
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from spurfies_amd import ops  # noqa: E402
from spurfies_amd import synthetic as syn  # noqa: E402
from tools.strong_proxy import build  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.set_num_threads(1)
ops.set_geo_mode("split_w")
scene = syn.make_scene(10000, seed=0, prior="fitted")
batches = bench.make_batches(scene, 32, rays, 0, 1, dev)
torch.manual_seed(1)
model, step = build(scene, dev, sync_free=True, use_graph=True, fork=False)
for i in range(20):
    step(*batches[i % 32])
torch.cuda.synchronize()
import gc

gc.collect()
gc.disable()
t0 = time.perf_counter()
for i in range(300):
    step(*batches[i % 32])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"300 steps: host returned after {t_host / 300 * 1e3:.3f} ms/step, all done after {t_all / 300 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    step(*batches[i % 32])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
