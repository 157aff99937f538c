import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np, torch
import runner
flat = runner.parse_overrides(["testlist=scan24", "vol=dtu_pn", "opt_stepNs=[30,0,0]", "vol.train.checkpoint_freq=0"])
args = runner.nest(flat)
from spurfies_amd import synthetic as syn
from spurfies_amd.train import VolOpt
scene = syn.make_scene(10000, seed=0, prior="fitted")
prior = {k: torch.from_numpy(np.asarray(v)) for k, v in scene["state"].items() if k.startswith(("F_geometry", "T."))}
t = VolOpt(args=args, batch_size=1, scan="scan24", root="/tmp/x", scene=scene, neural_points={"pts": scene["state"]["neural_pts"], "colors": scene["colors"]},
           prior_state_dict=prior, device="cuda", sync_free=True)
t.gen_dataset(0)
t.train_dataset.change_sampling_idx(t.num_pixels)
it = iter(t.train_dataloader)
b = next(it)
for _ in range(5):
    t.train_step(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    for b in t.train_dataloader:
        t.train_step(b)
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 60 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
