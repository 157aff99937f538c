"""Accuracy of the colour trunk's arithmetic modes against a float64 autograd evaluation of the same layers on the same pairs:
    python tools/color_accuracy.py [--gscale S]   (S multiplies the upstream gradient: the backward's values scale with 1 / (3 R))
per mode: agg3 (forward, per point), the colour-latent gradient and the weight gradients of F_color.0 / 2 / 4, relative to the tensor's max |.|."""
import argparse, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from spurfies_amd import ops, synthetic as syn
from spurfies_amd.torch_knnquery import VoxelGrid

ap = argparse.ArgumentParser(); ap.add_argument("--out", default="gpurun_out/color_accuracy.json"); ap.add_argument("--gscale", type=float, default=1.0 / 3072.0)
a = ap.parse_args()
scene = syn.make_scene(10000, seed=0)
dev = {k: torch.as_tensor(np.asarray(v)).float().cuda() for k, v in scene["state"].items()}
grid = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, scene["ranges"])
grid.set_pointset(dev["neural_pts"].unsqueeze(0))
packed = ops.pack_geometry_weights(dev)
rng = np.random.default_rng(0)
pts = scene["state"]["neural_pts"]
x = torch.from_numpy((pts[rng.integers(0, len(pts), 20000)] + rng.normal(0, 0.015, size=(20000, 3))).astype(np.float32)).cuda()
q = grid.query_dense(x.unsqueeze(1), 8, 2, 1)
ps, _, n = ops.compact_points(q["slot_valid"])
pl = ops.PairList(q["pidx"].reshape(-1, 8), ps, n)
P, NP = pl.host_counts()
geo = ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, with_grad=True)
wn = geo["wn"]
g_up = (torch.randn((P, 256), device="cuda") * a.gscale).contiguous()
names = [f"F_color.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]

# float64 reference
pp, po = pl.pair_point[:NP].long(), pl.pair_off.long()
slot = ps[:P].long()[pp]
j = torch.arange(NP, device="cuda") - po[pp]
idx = pl.nbr[slot, j].long()
T64 = dev["neural_feats_color"].double().requires_grad_(True)
W64 = [dev[k].double().requires_grad_(True) for k in names]
d = (x[slot] - dev["neural_pts"][idx]).double()
enc = [d] + [f(d * (2.0 ** l)) for l in range(6) for f in (torch.sin, torch.cos)]
h = torch.cat(enc + [T64[idx]], -1)
for li in range(3):
    h = torch.nn.functional.leaky_relu(h @ W64[2 * li].T + W64[2 * li + 1], 0.01)
agg64 = torch.zeros((P, 256), dtype=torch.float64, device="cuda").index_add_(0, pp, h * wn[:NP].double()[:, None])
grads64 = torch.autograd.grad((agg64 * g_up.double()).sum(), [T64] + W64)
ref = {"agg3": agg64.detach(), "latent": grads64[0], **{n_: g for n_, g in zip(names, grads64[1:])}}

res = {"pairs": NP, "points": P, "upstream_gradient_scale": a.gscale, "what": __doc__}
for mode, h2 in (("f32", {}), ("split", dict(color_fwd=False, color_bwd=False, wgrad=False)), ("h2", dict(color_fwd=True, color_bwd=True, wgrad=True)),
                 ("h2_fwd_only", dict(color_fwd=True, color_bwd=False, wgrad=False)), ("h2_bwd_only", dict(color_fwd=False, color_bwd=True, wgrad=False)),
                 ("h2_wgrad_only", dict(color_fwd=False, color_bwd=False, wgrad=True))):
    ops.set_color_mode("f32" if mode == "f32" else "split")
    ops.set_wgrad_mode("f32" if mode == "f32" else "split")
    prev = ops.set_h2(**h2)
    table = dev["neural_feats_color"].clone().requires_grad_(True)
    ws = [dev[k].clone().requires_grad_(True) for k in names]
    out = ops.ColorAgg.apply(table, *ws, x, wn, pl, dev["neural_pts"], P, NP)
    out.backward(g_up)
    got = {"agg3": out.detach().double(), "latent": table.grad.double(), **{n_: w.grad.double() for n_, w in zip(names, ws)}}
    res[mode] = {}
    for k, r in ref.items():
        e = (got[k] - r).abs()
        res[mode][k] = {"max_err_of_max": float(e.max() / r.abs().max()), "rms_err_of_rms": float(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt())}
    # rows of the latent gradient that sit on a LeakyReLU kink (a pre-activation within rounding of zero takes the other slope: the whole row moves)
    rn = ref["latent"].norm(dim=1)
    dev_rows = (got["latent"] - ref["latent"]).norm(dim=1) / rn.clamp(min=1e-30)
    touched = rn > 0
    res[mode]["latent_rows_off_by_more_than_1e-4_of_their_norm"] = int((dev_rows[touched] > 1e-4).sum())
    res[mode]["latent_rows_off_by_more_than_1e-5"] = int((dev_rows[touched] > 1e-5).sum())
    res[mode]["latent_rows_touched"] = int(touched.sum())
    res[mode]["latent_row_dev_median"] = float(dev_rows[touched].median())
    ops.set_h2(**prev)
    print(mode, {k: ((f"{v['max_err_of_max']:.2e}", f"{v['rms_err_of_rms']:.2e}") if isinstance(v, dict) else v) for k, v in res[mode].items()})
ops.set_color_mode("split"); ops.set_wgrad_mode("split")
json.dump(res, open(a.out, "w"), indent=1)
