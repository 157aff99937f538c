"""Optimisation step with the reference's recipe (spurfies/train.py:117-189, 330-397, 548-564):
frozen F_geometry/T, Adam over the remaining parameters (lr 5e-4; the reference's first param group
is empty), CosineAnnealingLR(T_max=100000, eta_min=3e-4), grad-norm clip 1.0, NaN/Inf gradient guard.

`TrainStep` is the part of `VolOpt.train_step` that touches the GPU; dataset / checkpoint / logging
plumbing around it is host code outside the hot path.  With `world_size > 1` rays are sharded
across ranks and gradients are summed with one all-reduce of a flat buffer (spurfies_amd/dist.py).
"""
from __future__ import annotations

import torch

from . import dist as sdist
from .model.loss import VolSDFLoss


def default_loss() -> VolSDFLoss:
    """Weights of config/ours.yaml:15-20."""
    return VolSDFLoss("torch.nn.L1Loss", local_weight=0.5, pseudo_weight=0.5, eikonal_weight=0.001, rgb_weight=1.0, tv_weight=0.01)


class TrainStep:
    def __init__(self, model, loss=None, lr=5.0e-4, grad_clip=True, process_group=None):
        self.model = model
        self.loss = loss or default_loss()
        model.freeze_prior()                                    # train.py:151-154
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = sdist.FlatGrads(self.params)                # .grad of every trainable tensor is a view of one buffer
        self.optimizer = torch.optim.Adam([{"params": [], "lr": 1e-2}, {"params": self.params, "lr": lr}])
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=100_000, eta_min=3e-4, last_epoch=-1)
        self.grad_clip = grad_clip
        self.group = process_group
        self.world = sdist.world_size(process_group)
        self.iter_step = 0
        self.skipped = 0

    def __call__(self, model_input, ground_truth):
        """model_input: {'intrinsics','uv','pose','local_data'} for THIS rank's rays; returns the loss dict."""
        self.model.train()
        model_input = dict(model_input)
        model_input["iter_step"] = self.iter_step
        out = self.model(model_input, fast=1)
        if self.world > 1:
            losses = sdist.sharded_loss(self.loss, out, ground_truth, self.group)
        else:
            losses = self.loss(out, ground_truth)
        self.flat.zero_()
        losses["loss"].backward()
        if self.world > 1:
            sdist.all_reduce_sum(self.flat.buffer, self.group)
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, 1.0)
        # train.py:548-564 — skip the update when a gradient is not finite (device-side test, no sync)
        finite = torch.isfinite(self.flat.buffer).all()
        self.flat.buffer.mul_(finite.to(self.flat.buffer.dtype))
        self.optimizer.step()
        self.scheduler.step()
        self.iter_step += 1
        return losses, out
