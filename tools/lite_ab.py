"""Evaluation chunk / image with and without the reduced-product sampler passes (SPF_ARITH_LITE), same box: python tools/lite_ab.py"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from spurfies_amd import ops, synthetic as syn
from spurfies_amd.conf import default_model_conf
from spurfies_amd.eval_graph import GraphedRenderer, ImageRenderer
from spurfies_amd.model.pointneus_disent import PointVolSDF

scene = syn.make_scene(10000, seed=0, prior="fitted")
st = scene["state"]
model = PointVolSDF(default_model_conf(near=0.5, grid_ranges=list(scene["ranges"])), 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]})
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
model.eval()
ops.set_geo_mode("split_w")
g = torch.Generator().manual_seed(777)
K = torch.from_numpy(scene["intrinsics"])[None].cuda()
batches = [{"intrinsics": K, "uv": torch.from_numpy(syn.make_pixels(1024, g))[None].cuda(), "pose": torch.from_numpy(scene["poses"][i % 3])[None].cuda()} for i in range(8)]
H, W = 576, 768
ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
uv_all = torch.stack([xs, ys], -1).reshape(1, -1, 2).float().cuda()
view = {"uv": uv_all, "pose": torch.from_numpy(scene["poses"][0])[None].cuda(), "intrinsics": K}
res = {}
for leg in ("full", "lite", "lite", "full"):
    model.sampler_lite = leg == "lite"
    render = GraphedRenderer(model, 1024, keys=("rgb_values", "depth_values", "normal_map"))
    with torch.no_grad():
        for i in range(30):
            render(batches[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            render(batches[i % 8])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 10
        it = model.ray_sampler.last_iters
        r = ImageRenderer(model, 1024, fast=-1, graph=True)
        r(dict(view, uv=uv_all[:, :4096]), 4096)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img = r(view, H * W)
        torch.cuda.synchronize(); ti = time.perf_counter() - t0
    res.setdefault(leg, []).append({"ms_per_chunk": ms, "iters": it, "image_s": ti, "rgb_mean": float(img["rgb_values"].mean())})
    print(leg, res[leg][-1], flush=True)
json.dump(res, open("gpurun_out/lite_ab.json", "w"), indent=1)
