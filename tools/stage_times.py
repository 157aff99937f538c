"""Where does a training step spend its time?  CUDA-event brackets around the stages of
PointVolSDF.forward / backward / optimiser (diagnostic; not the contract bench)."""
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


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    scene = syn.make_scene(n_points, seed=0)
    st = scene["state"]
    conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
    model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
    step = TrainStep(model)
    batches = bench.make_batches(scene, 8, 1024, 0, 1, torch.device("cuda"))
    for b in batches[:3]:
        step(*b)
    torch.cuda.synchronize()
    # wall-clock split of one step with syncs between phases
    for b in batches[3:6]:
        t = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.train()
        out = model(dict(b[0], iter_step=0), fast=1)
        torch.cuda.synchronize(); t["forward"] = time.perf_counter() - t0; t0 = time.perf_counter()
        losses = step.loss(out, b[1])
        step.flat.zero_()
        losses["loss"].backward()
        torch.cuda.synchronize(); t["loss+backward"] = time.perf_counter() - t0; t0 = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(step.params, 1.0)
        step.optimizer.step(); step.scheduler.step()
        torch.cuda.synchronize(); t["clip+adam"] = time.perf_counter() - t0
        print({k: round(v * 1e3, 2) for k, v in t.items()}, "P", model.stats)
    # forward sub-stages
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for b in batches[6:8]:
            step(*b)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
