"""Per-kernel micro-benchmarks on one MI355X (not the contract bench; see bench.py)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from spurfies_amd import ops, synthetic as syn  # noqa: E402
from spurfies_amd.torch_knnquery import VoxelGrid  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    scene = syn.make_scene(n_points, seed=0)
    dev = {k: torch.as_tensor(np.asarray(v)).float().cuda() for k, v in scene["state"].items()}
    grid = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, scene["ranges"])
    t0 = time.time()
    grid.set_pointset(dev["neural_pts"].unsqueeze(0))
    torch.cuda.synchronize()
    print("grid build s", time.time() - t0, grid.info())
    packed = ops.pack_geometry_weights(dev)
    rng = np.random.default_rng(0)
    pts = scene["state"]["neural_pts"]
    out = {}
    for n_q in (8192, 65536, 131072):
        x = torch.from_numpy((pts[rng.integers(0, len(pts), n_q)] + rng.normal(0, 0.015, size=(n_q, 3))).astype(np.float32)).cuda()
        q = grid.query_dense(x.unsqueeze(1), 8, 2, 1)
        ps, _, n = ops.compact_points(q["slot_valid"])
        nbr = q["pidx"].reshape(-1, 8)
        pl = ops.PairList(nbr, ps, n)
        P, pairs = pl.host_counts()
        t_knn = timeit(lambda: grid.query_dense(x.unsqueeze(1), 8, 2, 1))
        t_cmp = timeit(lambda: ops.compact_points(q["slot_valid"]))
        t_pl = timeit(lambda: ops.PairList(nbr, ps, n))
        t_f = timeit(lambda: ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, False))
        t_j = timeit(lambda: ops.geo_forward(x, pl, dev["neural_pts"], dev["neural_feats_geometry"], packed, 45.0, True))
        # algorithmic flops per padded row (8 per point): fwd 2*(35*256+3*256*256+256); jac adds 2*(3*256*256+256*35)
        f_fwd = 2.0 * (35 * 256 + 3 * 256 * 256 + 256)
        f_jac = 2.0 * (3 * 256 * 256 + 256 * 35)
        out[n_q] = dict(P=P, pairs=pairs, knn_ms=t_knn, compact_ms=t_cmp, pairs_ms=t_pl, geo_fwd_ms=t_f, geo_fwd_jac_ms=t_j,
                        fwd_tflops_pairs=pairs * f_fwd / t_f / 1e9, jac_tflops_pairs=pairs * (f_fwd + f_jac) / t_j / 1e9,
                        fwd_tflops_rows=P * 8 * f_fwd / t_f / 1e9)
        print(n_q, json.dumps(out[n_q]))
    # main-pass shape: 1024 rays x 98 samples, SR=80
    R, D = 1024, 98
    o = torch.tensor([2.2, 0.3, 0.4], device="cuda")
    tgt = torch.randn((R, 3), device="cuda") * 0.3
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    z = torch.sort(torch.rand((R, D), device="cuda") * 1.6 + 1.4, dim=1)[0]
    xr = o + z[..., None] * d[:, None]
    t_main = timeit(lambda: grid.query_dense(xr, 8, 2, 80))
    qq = grid.query_dense(xr, 8, 2, 80)
    print("main-pass query ms", t_main, "valid pts", int(qq["slot_valid"].sum()))


if __name__ == "__main__":
    main()
