"""Evaluation-mode throughput (reported separately from the train-step metric), on the bench scene:
  (1) rendering: PointVolSDF.forward(input, fast=-1) under torch.no_grad() — the full error-bounded sampler (up to 5 iterations),
      kNN, SDF + normals, colour, compositing — rays/s and ray-samples/s with the realised sampler iteration count;
  (2) mesh-extraction entry: get_sdf_eval over a dense grid of points — points/s.
Run on the GPU box from the repo root."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from spurfies_amd import synthetic as syn  # noqa: E402
from spurfies_amd.conf import default_model_conf  # noqa: E402
from spurfies_amd.model.pointneus_disent import PointVolSDF  # noqa: E402

dev = torch.device("cuda", 0)
scene = syn.make_scene(10000, seed=0, prior="fitted")
st = scene["state"]
conf = default_model_conf(near=0.5, grid_ranges=list(scene["ranges"]))
model = PointVolSDF(conf, 24, "dtu", neural_points={"pts": st["neural_pts"], "colors": scene["colors"]}, device=dev)
model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in st.items()}, strict=False)
model.eval()
g = torch.Generator().manual_seed(3)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = torch.from_numpy(scene["intrinsics"])[None].to(dev)
batches = [{"intrinsics": K, "uv": torch.from_numpy(syn.make_pixels(R, g))[None].to(dev),
            "pose": torch.from_numpy(scene["poses"][i % len(scene["poses"])])[None].to(dev), "local_data": None} for i in range(12)]
with torch.no_grad():
    for b in batches[:3]:
        out = model(dict(b), fast=-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[3:]:
        out = model(dict(b), fast=-1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 9
    iters = getattr(model.ray_sampler, "last_iters", None)
    print(f"render (eval, fast=-1): {R} rays in {dt * 1e3:.2f} ms = {R / dt / 1e3:.1f} k rays/s"
          + (f", sampler iterations of the last batch: {iters}" if iters is not None else ""))
    n = 128
    lin = torch.linspace(-0.7, 0.7, n, device=dev)
    pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    for _ in range(2):
        sdf = torch.cat([model.get_sdf_eval(c) for c in pts.split(100000)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sdf = torch.cat([model.get_sdf_eval(c) for c in pts.split(100000)])      # the reference's chunking (pointneus_disent.py:249-298)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hit = float((sdf < 999.0).float().mean())
    print(f"get_sdf_eval: {pts.shape[0]} grid points ({n}^3, {100 * hit:.1f} % near the cloud) in {dt * 1e3:.1f} ms = {pts.shape[0] / dt / 1e6:.1f} M points/s")

    # the reference's mesh extraction samples ~512 points along the shortest axis of the box (eval_spurfies.py:141-157, plots.py:188-287)
    if "--sweep512" in sys.argv:
        from spurfies_amd.utils import surface

        b = scene["base_radius"] * 1.25
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        verts, faces, vol, grid = surface.extract_surface(model.get_sdf_eval, 512, [-b] * 3, [b] * 3)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        vol = surface.sdf_volume(model.get_sdf_eval, grid)
        torch.cuda.synchronize()
        t_sweep = time.perf_counter() - t0
        print(f"512-grid sweep: {vol.size} points in {t_sweep:.2f} s = {vol.size / t_sweep / 1e6:.1f} M points/s (incl. H2D of the chunks and D2H of the "
              f"volume); with the host-side triangulation + largest component: {t_all:.1f} s, {0 if faces is None else len(faces)} faces")
