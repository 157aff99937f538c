"""Is the eager sync-free step GPU-bound?  Host time to ENQUEUE a step (no synchronisation) vs the GPU's time to run it."""
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

scene = syn.make_scene(10000, seed=0)
st = scene["state"]
conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
step = TrainStep(model, sync_free=True)
batches = bench.make_batches(scene, 30, 1024, 0, 1, torch.device("cuda"))
for b in batches[:5]:
    step(*b)
torch.cuda.synchronize()
enq = []
t_all = time.perf_counter()
for b in batches[5:30]:
    t0 = time.perf_counter()
    step(*b)
    enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
# the first few enqueues run ahead of an idle GPU: they are the pure host cost; later ones block on the launch queue depth
print(f"host enqueue, first 3 steps: {[round(1e3 * e, 2) for e in enq[:3]]} ms;  median of all: {1e3 * sorted(enq)[len(enq) // 2]:.2f} ms;  "
      f"wall per step: {1e3 * t_all / len(enq):.2f} ms")
