"""Which torch ops (with shapes) still cost GPU time in the sync-free step?"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for b in batches[3:5]:
        step(*b)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:28]:
    print(f"{e.key:28s} n={e.count:4d} gpu_us={e.device_time_total/2:9.1f}  {str(e.input_shapes)[:110]}")
