"""Accuracy of the geometry kernel's arithmetic modes against a float64 evaluation of the same MLP on the same pairs (main-pass-shaped input):
    python tools/engine_accuracy.py [--out gpurun_out/engine_accuracy.json]
per mode: relative error of the per-pair SDF (vs float64), of d sdf / d x and of the latent Jacobian rows; the f32 line IS what plain fp32 arithmetic gives."""
import argparse, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from spurfies_amd import ops
from tests.test_gpu_geo import _setup

ap = argparse.ArgumentParser(); ap.add_argument("--out", default="gpurun_out/engine_accuracy.json"); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
scene, st, cfg, x, dev, grid, packed = _setup(n_points=10000, n_query=40000, seed=6)
xt = torch.from_numpy(x).cuda()
q = grid.query_dense(xt.unsqueeze(1), cfg.k, cfg.r, 1)
ps, _, n = ops.compact_points(q["slot_valid"])
pl = ops.PairList(q["pidx"].reshape(-1, cfg.k), ps, n)
P, NP = (int(v) for v in pl.counts.tolist())
# float64 reference of the per-pair MLP: rows in the kernels' pair order
pp, po = pl.pair_point[:NP].long(), pl.pair_off.long()
slot = ps[:P].long()[pp]
j = torch.arange(NP, device="cuda") - po[pp]
idx = pl.nbr[slot, j].long()
lat = dev["neural_feats_geometry"][idx].double()
d = (xt[slot] - dev["neural_pts"][idx]).double()
inp = torch.cat([lat, d], -1).requires_grad_(True)
W = lambda k: dev[k].double()
h = inp
for li in (0, 2, 4, 6):
    h = torch.nn.functional.leaky_relu(h @ W(f"F_geometry.{li}.weight").T + W(f"F_geometry.{li}.bias"), 0.01)
h = h @ W("F_geometry.8.weight").T + W("F_geometry.8.bias")
sdf64 = (h @ W("T.0.weight").T + W("T.0.bias")).squeeze(-1)
(jac64,) = torch.autograd.grad(sdf64.sum(), inp)
res = {"pairs": NP, "what": __doc__}
for mode in ("f32", "split", "split_w", "h2"):
    ops.set_geo_mode(mode)
    out = ops.geo_forward(xt, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, cfg.rbf, with_grad=True)
    tmp = out["pair_tmp"][:NP].double()
    sdf_j, gx = tmp[:, 1], tmp[:, 2:5]
    jl = out["jac"][:NP].double()
    e_s = (sdf_j - sdf64.detach()).abs() / sdf64.detach().abs().clamp(min=1e-3)
    scale_x = jac64[:, 32:].abs().max()
    scale_l = jac64[:, :32].abs().max()
    e_x = (gx - jac64[:, 32:]).abs().max(-1).values / scale_x
    e_l = (jl - jac64[:, :32]).abs().max(-1).values / scale_l
    kink = (e_x > 1e-3) | (e_l > 1e-3)          # a LeakyReLU unit within rounding of zero takes the other slope: excluded from the statistics, counted
    res[mode] = {"sdf_rel_err_rms": float(e_s.pow(2).mean().sqrt()), "sdf_rel_err_max": float(e_s.max()), "sdf_rel_err_p999": float(e_s.quantile(0.999)),
                 "dx_err_rms_of_scale": float(e_x[~kink].pow(2).mean().sqrt()), "dx_err_max_of_scale": float(e_x[~kink].max()),
                 "jac_err_rms_of_scale": float(e_l[~kink].pow(2).mean().sqrt()), "jac_err_max_of_scale": float(e_l[~kink].max()), "kink_rows": int(kink.sum())}
    print(mode, res[mode])
ops.set_geo_mode("h2")
json.dump(res, open(a.out, "w"), indent=1)
