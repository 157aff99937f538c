import sys
sys.path.insert(0, ".")
import numpy as np, torch
from tests.helpers import inputs_of, load_golden, scene_of
from tests.test_gpu_model import build_model
fx = load_golden("step_eval_r24.npz")
scene = scene_of(fx)
model = build_model(scene, train=False)
inp = inputs_of(fx, scene, device="cuda")
torch.manual_seed(int(fx["meta.seed"]) + 7)
out = model(inp, fast=-1)
for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz", "normal_map"):
    a, b = out[k].detach().cpu().numpy(), fx[f"out.{k}"]
    err = np.abs(a - b)
    print(k, "max abs %.2e" % err.max(), "max rel(+1e-4) %.2e" % (err / (np.abs(b) + 1e-4)).max(), "frac > 1e-4+1e-4|b|: %.4f" % (err > 1e-4 + 1e-4 * np.abs(b)).mean())
for name in ("step_train_r128.npz", "step_train_far.npz"):
    fx = load_golden(name); scene = scene_of(fx); model = build_model(scene); inp = inputs_of(fx, scene, device="cuda")
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    out = model(inp, fast=1)
    for k in ("rgb_values", "depth_values", "depth_vals", "weights", "xyz"):
        a, b = out[k].detach().cpu().numpy(), fx[f"out.{k}"]
        err = np.abs(a - b)
        print(name, k, "max abs %.2e" % err.max(), "frac > 1e-4+1e-4|b|: %.5f" % (err > 1e-4 + 1e-4 * np.abs(b)).mean())
# ---- the merged evaluation image (reference evaluation configuration, near = 0.0)
from spurfies_amd.eval_graph import ImageRenderer
fx = load_golden("eval_image_near0.npz")
scene = scene_of(fx)
model = build_model(scene, train=False, near=float(fx["meta.near"]))
inp = inputs_of(fx, scene, device="cuda")
total, chunk = fx["in.uv"].shape[0], int(fx["meta.chunk"])
for graph in (False, True):
    r = ImageRenderer(model, chunk, fast=-1, graph=graph, keep_weights=True)
    torch.manual_seed(int(fx["meta.seed"]) + 7)
    out = r(inp, total)
    torch.cuda.synchronize()
    for k in ("rgb_values", "depth_values", "weights", "normal_map"):
        a, b = out[k].cpu().numpy().reshape(total, -1), fx[f"out.{k}"].reshape(total, -1)
        err = np.abs(a - b)
        rowbad = (err > 2e-5 + 1e-4 * np.abs(b)).any(axis=1)
        print("image graph=%s" % graph, k, "max abs %.2e" % err.max(), "pixels outside 1e-4/2e-5: %d of %d" % (rowbad.sum(), total), "99.9th pct abs %.2e" % np.percentile(err, 99.9))
