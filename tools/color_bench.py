"""Colour trunk / head kernels alone: forward with and without the training stores, backward, on main-pass-shaped inputs."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tools.microbench import timeit  # noqa: E402
from spurfies_amd import ops, synthetic as syn  # noqa: E402
from spurfies_amd.torch_knnquery import VoxelGrid  # noqa: E402

if len(sys.argv) > 1:
    ops.set_color_mode(sys.argv[1])          # split | f32
RAYS = len(sys.argv) > 2 and sys.argv[2] == "rays"      # queries along rays (neighbour sets overlap from sample to sample, as in a step)
scene = syn.make_scene(10000, seed=0)
dev = {k: torch.as_tensor(np.asarray(v)).float().cuda() for k, v in scene["state"].items()}
grid = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, scene["ranges"])
grid.set_pointset(dev["neural_pts"].unsqueeze(0))
packed = ops.pack_geometry_weights(dev)
rng = np.random.default_rng(0)
pts = scene["state"]["neural_pts"]
n_q = 56000
if RAYS:
    n_r, n_s = 1000, 56
    o = pts[rng.integers(0, len(pts), n_r)] + rng.normal(0, 0.01, size=(n_r, 3))
    d = rng.normal(size=(n_r, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = (np.arange(n_s) - n_s / 2) * 0.0015
    x = torch.from_numpy((o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3).astype(np.float32)).cuda()
else:
    x = torch.from_numpy((pts[rng.integers(0, len(pts), n_q)] + rng.normal(0, 0.015, size=(n_q, 3))).astype(np.float32)).cuda()
q = grid.query_dense(x.unsqueeze(1), 8, 2, 1)
ps, _, n = ops.compact_points(q["slot_valid"])
pl = ops.PairList(q["pidx"].reshape(-1, 8), ps, n)
P, NP = pl.host_counts()
geo = ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, with_grad=True)
fcp = [dev[f"F_color.{i}.{w}"].clone().requires_grad_(True) for i in (0, 2, 4) for w in ("weight", "bias")]
table = dev["neural_feats_color"].clone().requires_grad_(True)
f_fwd = 2.0 * (103 * 256 + 2 * 256 * 256)
f_bwd = 2.0 * (2 * 256 * 256 + 256 * 64)


def fwd_train():
    return ops.ColorAgg.apply(table, *fcp, x, geo["wn"], pl, dev["neural_pts"], P, NP)


def fwd_eval():
    with torch.no_grad():
        return ops.ColorAgg.apply(table, *fcp, x, geo["wn"], pl, dev["neural_pts"], P, NP)


t_e, t_t = timeit(fwd_eval), timeit(fwd_train)
out = fwd_train()
g = torch.randn_like(out)
ctx_fn = out.grad_fn


def bwd_only():
    # the HIP backward kernel alone (weight-gradient GEMMs excluded): call the C entry point through the saved context
    wn, pk, act0, act1, act2, masks = ctx_fn.saved_tensors
    rows = act1.shape[0]
    G = [torch.empty((rows, 256), device="cuda") for _ in range(3)]
    gb = torch.zeros((3, 256), device="cuda")
    gf = torch.zeros((table.shape[0], 64), device="cuda")
    from spurfies_amd import _lib
    _lib.check(_lib.lib().spf_color_backward(_lib.ptr(g), _lib.ptr(pl.nbr), _lib.ptr(wn), _lib.ptr(pl.point_slot), _lib.ptr(pl.pair_off),
                                             _lib.ptr(pl.pair_point), _lib.ptr(pl.n_pairs), NP, pl.k, _lib.ptr(pk), _lib.ptr(masks), _lib.ptr(G[0]),
                                             _lib.ptr(G[1]), _lib.ptr(G[2]), _lib.ptr(gb[0]), _lib.ptr(gb[1]), _lib.ptr(gb[2]), _lib.ptr(gf), None,
                                             ops._arith_of(ops._ARITH["color"], "color_bwd"), _lib.stream_ptr()), "bwd")


t_b = timeit(bwd_only)
print(f"[{ops.color_mode()}] P={P} pairs={NP}  fwd eval {t_e:.3f} ms ({NP*f_fwd/t_e/1e9:.1f} TF)  fwd train {t_t:.3f} ms ({NP*f_fwd/t_t/1e9:.1f} TF)  "
      f"bwd kernel {t_b:.3f} ms ({NP*f_bwd/t_b/1e9:.1f} TF)")
