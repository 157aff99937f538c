"""Debug aid: layer-4 LeakyReLU sign words written by color_forward_x3_kernel against signs recomputed from its stored act2 tiles."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import tools.color_bench as cb  # noqa: E402

out = cb.fwd_train()
wn, pk, act0, act1, act2, masks = out.grad_fn.saved_tensors
tiles = masks.shape[0]
w4, b4 = cb.fcp[4].detach(), cb.fcp[5].detach()
a2 = act2.view(tiles * 4, 256, 16).transpose(1, 2).reshape(-1, 256)          # K-major 16-row blocks -> rows
h3 = a2 @ w4.t() + b4
NP = cb.NP
m = masks.view(tiles, 3, 512)[:, 2, :].reshape(tiles * 64, 8)          # [row][8 words]
bits = ((m[:, :, None] >> torch.arange(32, device="cuda")[None, None, :]) & 1).reshape(-1, 256).bool()
sure = h3.abs() > 1e-4
bad = ((bits != (h3 > 0)) & sure)[:NP]
print("rows", NP, "wrong sign bits", int(bad.sum()), "of", int(sure[:NP].sum()))
if bad.any():
    r, c = torch.nonzero(bad)[0].tolist()
    print("first bad: row", r, "in-tile row", r % 64, "feature", c, "h3", float(h3[r, c]))
    print("bad per in-tile row:", torch.bincount(torch.nonzero(bad)[:, 0] % 64, minlength=64).tolist())
    print("bad per feature block of 32:", torch.bincount(torch.nonzero(bad)[:, 1] // 32, minlength=8).tolist())
