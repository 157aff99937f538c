"""Host-side (CPU) cost of one training step by phase — are we launch-bound?"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from spurfies_amd import synthetic as syn  # noqa: E402
from spurfies_amd.conf import default_model_conf  # noqa: E402
from spurfies_amd.model.pointneus_disent import PointVolSDF  # noqa: E402
from spurfies_amd.train import TrainStep  # noqa: E402

scene = syn.make_scene(10000, seed=0, prior="fitted")
st = scene["state"]
conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
step = TrainStep(model, sync_free="--default" not in sys.argv)
batches = bench.make_batches(scene, 10, 1024, 0, 1, torch.device("cuda"))
for b in batches[:4]:
    step(*b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for b in batches[4:8]:
        step(*b)
    torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=50))
n_launch = sum(e.count for e in ka if e.key.startswith("hipLaunchKernel") or e.key.startswith("hipExtModuleLaunch") or "LaunchKernel" in e.key)
print("kernel launches per step ~", n_launch / 4)
# pure-python wall for phases without GPU sync inside (enqueue cost only)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches[8:10]:
    step(*b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue ms/step", (t1 - t0) / 2 * 1e3, "drain ms", (t2 - t1) * 1e3)
