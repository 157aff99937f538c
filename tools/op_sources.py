"""Which source lines of the host code issue the small torch kernels of the sync-free step? (GPU time per step by line)"""
import collections
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
from spurfies_amd import synthetic as syn  # noqa: E402
from spurfies_amd.conf import default_model_conf  # noqa: E402
from spurfies_amd.model.pointneus_disent import PointVolSDF  # noqa: E402
from spurfies_amd.train import TrainStep  # noqa: E402
from torch.profiler import profile, ProfilerActivity

scene = syn.make_scene(10000, seed=0)
st = scene["state"]
conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
step = TrainStep(model, sync_free=True)
batches = bench.make_batches(scene, 6, 1024, 0, 1, torch.device("cuda"))
for b in batches[:3]:
    step(*b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for b in batches[3:5]:
        step(*b)
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    where = "?"
    for fr in e.stack:
        if "spurfies_amd" in fr or "bench.py" in fr:
            where = fr.split("spurfies_amd/")[-1][:70]
            break
    k = (where, e.name)
    acc[k][0] += 1
    acc[k][1] += e.device_time_total
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total aten GPU us/step {tot / 2:.1f}")
for (w, n), (c, t) in rows[:70]:
    print(f"{t / 2:8.1f} us  n={c / 2:5.1f}  {n:24s} {w}")
