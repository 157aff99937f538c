"""Where do the few-per-cent pixel deviations of the evaluation-image fixture come from?  Sampler depths z of one chunk: HIP vs the CPU oracle
(which reproduces the reference on this fixture to 2e-4).   python3 tools/image_z_check.py [chunk]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import path as P
from tests.helpers import inputs_of, load_golden, scene_of
from tests.test_gpu_model import build_model

fx = load_golden("eval_image_near0.npz")
scene = scene_of(fx)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
chunk = int(fx["meta.chunk"])
lo, hi = c * chunk, min((c + 1) * chunk, fx["in.uv"].shape[0])
uv = torch.from_numpy(fx["in.uv"])[None, lo:hi]
base = {"intrinsics": torch.from_numpy(scene["intrinsics"])[None], "pose": torch.from_numpy(scene["poses"][int(fx["meta.view"])])[None]}
st = P.load_state(scene["state"], requires_grad=False)
cfg = P.PathConfig(ranges=tuple(scene["ranges"]), near=float(fx["meta.near"]))
stages = {}
torch.set_num_threads(8)
with torch.enable_grad():
    oout = P.forward(dict(base, uv=uv), st, cfg, training=False, fast=-1, stages=stages)
zo = stages["z"].detach().numpy()
model = build_model(scene, train=False, near=float(fx["meta.near"]))
model.keep_stages = True
inp = {k: v.cuda() for k, v in dict(base, uv=uv).items()}
with torch.no_grad():
    out = model(dict(inp, local_data=None), fast=-1)
zh = None
pts = model.ray_sampler.last_points            # [R, M, 3] = o + z d
cam = model.stages.get("cam_loc") if hasattr(model, "stages") and model.stages else None
# z from the points: project on the ray direction
dirs_b, cam_b = P.camera_rays(uv, base["pose"], base["intrinsics"])
d = dirs_b.reshape(-1, 3).numpy(); o = cam_b.reshape(1, 3).numpy()
zh = ((pts.cpu().numpy() - o[None]) * d[:, None, :]).sum(-1)
dz = np.abs(zh - zo).max(axis=1)
rgb_err = np.abs(out["rgb_values"].cpu().numpy() - fx["out.rgb_values"][lo:hi]).max(axis=1)
rgb_err_o = np.abs(oout["rgb_values"].detach().numpy() - fx["out.rgb_values"][lo:hi]).max(axis=1)
print("chunk", c, "rays", hi - lo, "iters hip", model.ray_sampler.last_iters, "oracle", stages["trace"].get("iters"))
print("oracle vs reference rgb: max %.2e" % rgb_err_o.max())
for thr in (1e-5, 1e-4, 1e-3, 1e-2):
    print("rays with max |z_hip - z_oracle| > %.0e: %d;  of them with rgb err > 1e-4: %d" % (thr, (dz > thr).sum(), ((dz > thr) & (rgb_err > 1e-4)).sum()))
print("rays with rgb err > 1e-4: %d, of them with dz <= 1e-5: %d" % ((rgb_err > 1e-4).sum(), ((rgb_err > 1e-4) & (dz <= 1e-5)).sum()))
bad = np.argsort(-rgb_err)[:5]
for r in bad:
    k = int(np.abs(zh[r] - zo[r]).argmax())
    print("ray", r, "rgb err %.2e" % rgb_err[r], "max dz %.2e at sample %d (z_o %.5f z_h %.5f)" % (dz[r], k, zo[r, k], zh[r, k]), "n differing samples (>1e-5):", int((np.abs(zh[r] - zo[r]) > 1e-5).sum()))
