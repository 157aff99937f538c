"""bf16-piece geometry kernel (ops.set_geo_mode('split')) against the fp32-MFMA kernel: outputs and speed on main-pass-shaped input."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tools.microbench import timeit  # noqa: E402
from spurfies_amd import ops, synthetic as syn  # noqa: E402
from spurfies_amd.torch_knnquery import VoxelGrid  # noqa: E402

scene = syn.make_scene(10000, seed=0)
dev = {k: torch.as_tensor(np.asarray(v)).float().cuda() for k, v in scene["state"].items()}
grid = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, scene["ranges"])
grid.set_pointset(dev["neural_pts"].unsqueeze(0))
packed = ops.pack_geometry_weights(dev)
rng = np.random.default_rng(0)
pts = scene["state"]["neural_pts"]
n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 56000
x = torch.from_numpy((pts[rng.integers(0, len(pts), n_q)] + rng.normal(0, 0.015, size=(n_q, 3))).astype(np.float32)).cuda()
q = grid.query_dense(x.unsqueeze(1), 8, 2, 1)
ps, _, n = ops.compact_points(q["slot_valid"])
pl = ops.PairList(q["pidx"].reshape(-1, 8), ps, n)
P, NP = pl.host_counts()
f_all = 2.0 * (35 * 256 + 3 * 256 * 256 + 256) + 2.0 * (3 * 256 * 256 + 256 * 35)
res = {}
for mode in ("f32", "split"):
    ops.set_geo_mode(mode)
    out = ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, with_grad=True)
    res[mode] = {k: out[k].clone() for k in ("sdf", "grad", "wn", "jac")}
    t_j = timeit(lambda: ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, True))
    t_f = timeit(lambda: ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, False))
    print(f"{mode:6s} P={P} pairs={NP}  fwd {t_f:.3f} ms  fwd+jac {t_j:.3f} ms ({NP * f_all / t_j / 1e9:.1f} fp32-equivalent TFLOP/s)")
ops.set_geo_mode("f32")
for k in ("sdf", "grad", "wn", "jac"):
    a, b = res["f32"][k], res["split"][k]
    print(f"{k:5s} max |diff| {float((a - b).abs().max()):.3e}   max |value| {float(a.abs().max()):.3e}")
