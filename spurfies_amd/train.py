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
    def __init__(self, model, loss=None, lr=5.0e-4, grad_clip=True, process_group=None, sync_free=False, use_graph=False):
        """sync_free: static shapes and device-side counts everywhere — no host synchronisation inside the step (the
        default path reads [P, n_pairs] back once per step to size the colour buffers exactly).
        use_graph (implies sync_free): forward + loss + backward (~230 kernel launches) are captured once into a hipGraph
        and replayed; the gradient all-reduce, clipping and Adam stay eager."""
        self.model = model
        sync_free = sync_free or use_graph
        self.sync_free = sync_free
        self.use_graph = use_graph
        self._graph = None
        model.sync_free = sync_free
        self._draws = None
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
        """model_input: {'intrinsics','uv','pose','local_data'} for THIS rank's rays; returns (loss dict, model outputs)."""
        self.model.train()
        if self.use_graph:
            losses, out = self._graphed_forward_backward(model_input, ground_truth)
        else:
            losses, out = self._forward_backward(model_input, ground_truth)
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

    def _forward_backward(self, model_input, ground_truth):
        model_input = dict(model_input)
        model_input["iter_step"] = self.iter_step
        if self.sync_free:
            self._refresh_draws(model_input["uv"].shape[1], model_input["uv"].device)
        out = self.model(model_input, fast=1)
        if self.world > 1:
            losses = sdist.sharded_loss(self.loss, out, ground_truth, self.group)
        else:
            losses = self.loss(out, ground_truth)
        self.flat.zero_()
        losses["loss"].backward()
        return losses, out

    # ------------------------------------------------------------------ hipGraph path
    def _graphed_forward_backward(self, model_input, ground_truth):
        dev = model_input["uv"].device
        keys_in = ("intrinsics", "uv", "pose")
        if self._graph is None or self._static_in["uv"].shape != model_input["uv"].shape:
            if self.world > 1:
                raise NotImplementedError("use_graph with ray sharding: the count all-reduce inside the loss is not captured yet")
            self._static_in = {k: model_input[k].clone() for k in keys_in}
            self._static_gt = {k: ground_truth[k].to(dev).clone() for k in ("rgb", "mask")}
            # warm-up on a side stream (builds the cell table, TV graph, workspaces, allocator pools) without touching the
            # training trajectory: parameters are restored afterwards and no optimiser step is taken
            rng = torch.get_rng_state()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._forward_backward(dict(self._static_in, local_data=None), self._static_gt)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.set_rng_state(rng)
            self._refresh_draws(model_input["uv"].shape[1], dev)      # allocates the persistent draw buffers
            torch.set_rng_state(rng)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                out = self.model(dict(self._static_in, local_data=None, iter_step=0), fast=1)
                losses = self.loss(out, self._static_gt)
                self.flat.zero_()
                losses["loss"].backward()
            self._static_out = (losses, out)
        for k in keys_in:
            self._static_in[k].copy_(model_input[k], non_blocking=True)
        for k in ("rgb", "mask"):
            self._static_gt[k].copy_(ground_truth[k], non_blocking=True)
        self._refresh_draws(model_input["uv"].shape[1], dev)
        self._graph.replay()
        return self._static_out

    def _refresh_draws(self, R, dev):
        """The reference draws its random numbers from the CPU generator and moves them (ray_sampler.py:55,514,550,562); in
        sync-free mode the same calls are issued here, in the same order, into persistent device buffers the sampler reads."""
        s = self.model.ray_sampler
        n0, N, Ne, M = s.N_samples_eval, s.N_samples, s.N_samples_extra, s.N_samples + 2 + s.N_samples_extra
        if self._draws is None or self._draws["t_rand"].shape[0] != R:
            self._draws = {"t_rand": torch.empty((R, n0), device=dev), "u": torch.empty((R, N), device=dev),
                           "sel": torch.empty((Ne,), dtype=torch.int32, device=dev)}
            self._pinned = {"t_rand": torch.empty((R, n0)).pin_memory(), "u": torch.empty((R, N)).pin_memory(),
                            "sel": torch.empty((Ne,), dtype=torch.int32).pin_memory()}
        torch.rand((R, n0), out=self._pinned["t_rand"])
        torch.rand((R, N), out=self._pinned["u"])
        self._pinned["sel"].copy_(torch.randperm(n0)[:Ne])
        torch.randint(M, (R,))                       # the unused eikonal index (:562) — keeps the generator in step
        for k, v in self._draws.items():
            v.copy_(self._pinned[k], non_blocking=True)
        s.draws = self._draws
