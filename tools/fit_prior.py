"""Fit a synthetic local geometry prior (F_geometry + T) so that the RBF-blended SDF of a synthetic scene is the signed
distance to its analytic surface ("a variant fitted so that SDF ~ signed distance ... to get realistic
sampler convergence / occupancy").  The reference's own prior, ckpt/local_prior.pt, is a separate download.

Convention: a neural point's geometry latent carries its surface normal, g[:3] = 0.5 n (remaining dims: small noise the
network learns to ignore); the per-pair target is the local tangent-plane distance n . x_pi, so the blended SDF is the
implicit-moving-least-squares surface of the oriented cloud.

    python tools/fit_prior.py            # ~5 min on 8 cores; writes spurfies_amd/data/prior_fitted.npz (fp32, ~1 MB)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "spurfies_amd", "data", "prior_fitted.npz")


def main(steps=6000, batch=8192, seed=0):
    torch.manual_seed(seed)
    torch.set_num_threads(8)
    lrelu = lambda: torch.nn.LeakyReLU(inplace=True)
    F = torch.nn.Sequential(torch.nn.Linear(35, 256), lrelu(), torch.nn.Linear(256, 256), lrelu(), torch.nn.Linear(256, 256), lrelu(),
                            torch.nn.Linear(256, 256), lrelu(), torch.nn.Linear(256, 256))
    T = torch.nn.Sequential(torch.nn.Linear(256, 1))
    params = list(F.parameters()) + list(T.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=1e-5)
    g = torch.Generator().manual_seed(seed + 1)
    t0 = time.time()
    for it in range(steps):
        n = torch.nn.functional.normalize(torch.randn((batch, 3), generator=g), dim=-1)
        lat = 0.05 * torch.randn((batch, 32), generator=g)
        lat[:, :3] = 0.5 * n + 0.01 * torch.randn((batch, 3), generator=g)
        # x_pi inside the kNN radius (0.05) with margin, denser near the centre like real neighbour offsets
        x = torch.nn.functional.normalize(torch.randn((batch, 3), generator=g), dim=-1) * 0.075 * torch.rand((batch, 1), generator=g) ** (1 / 2)
        target = (2.0 * lat[:, :3] * x).sum(-1, keepdim=True)
        pred = T(F(torch.cat([lat, x], -1)))
        loss = ((pred - target) ** 2).mean() * 1e4 + (pred - target).abs().mean() * 1e2
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if it % 500 == 0 or it == steps - 1:
            print(it, f"rmse {float(((pred - target) ** 2).mean().sqrt()):.2e} mae {float((pred - target).abs().mean()):.2e} {time.time() - t0:.0f}s", flush=True)
    sd = {f"F_geometry.{2 * i}.{k}": getattr(F[2 * i], k).detach().numpy().astype(np.float32) for i in range(5) for k in ("weight", "bias")}
    sd.update({"T.0.weight": T[0].weight.detach().numpy().astype(np.float32), "T.0.bias": T[0].bias.detach().numpy().astype(np.float32)})
    np.savez_compressed(OUT, **sd)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
