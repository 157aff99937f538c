"""Multi-GPU: ray sharding + one gradient all-reduce per step.

The reference is single-GPU; rays are independent given replicated state, so each rank renders a
strided slice of the batch with a full replica of cloud / latents / MLPs and the ranks exchange
exactly one flat fp32 buffer per step (RCCL over xGMI when the backend is "nccl", gloo in CPU
tests) plus one tiny count vector so that mean-type losses are normalised by GLOBAL counts — the
sum of the ranks' losses is then the single-GPU batch loss, and the summed gradient is its gradient.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_rays(n_rays: int, group=None) -> torch.Tensor:
    """Indices of this rank's rays: every rank draws the same permutation (same seed) and takes a
    strided slice."""
    return torch.arange(rank(group), n_rays, world_size(group))


def all_reduce_sum(t: torch.Tensor, group=None, force=False):
    """force: issue the collective even on a communicator of one (a sum over one rank: the values stay) — how a one-GPU box exercises the
    step's collectives on RCCL's own stream (TrainStep(force_collectives=True))."""
    if world_size(group) > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_model(model, optimizer=None, group=None, src=0):
    """Make every rank's replica identical to rank `src`'s: the trainable parameters (one broadcast of FlatAdam's flat buffer
    when there is one), the frozen prior weights and the buffers (the cloud)."""
    if world_size(group) == 1:
        return
    flat = getattr(optimizer, "_flat", None) if optimizer is not None else None
    done = set()
    if flat is not None:
        dist.broadcast(flat["param"], src=src, group=group)
        done = {id(p) for g in optimizer.param_groups for p in g["params"]}
    for p in model.parameters():
        if id(p) not in done:
            dist.broadcast(p.data, src=src, group=group)
            torch.autograd.graph.increment_version(p)
    for p in model.parameters():
        if id(p) in done:
            torch.autograd.graph.increment_version(p)
    for b in model.buffers():
        dist.broadcast(b, src=src, group=group)


class FlatGrads:
    """One contiguous fp32 buffer holding every trainable tensor's gradient ([N*64 + N*32 + F_color +
    R + beta] floats); each parameter's .grad is a view into it, so the all-reduce needs no packing."""

    def __init__(self, params):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.buffer = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.buffer[off: off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.buffer.zero_()
        off = 0
        for p in self.params:  # re-attach in case an optimizer / zero_grad(set_to_none) dropped the views
            if p.grad is None or p.grad.data_ptr() != self.buffer.data_ptr() + 4 * off:
                p.grad = self.buffer[off: off + p.numel()].view_as(p)
                if getattr(p, "_spf_grad_sink", None) is not None:
                    p._spf_grad_sink = p.grad
            off += p.numel()


class BucketedAllReduce:
    """The flat gradient buffer as named BUCKETS (contiguous ranges), each all-reduced asynchronously as soon as its last producer kernel
    has been enqueued — instead of one dense all-reduce behind the whole backward (round-2 verdict item 8).

    `ready(name)` is called by the backward code (ops.set_bucket_hook) right after the launch that completes a bucket: with the "nccl" (RCCL)
    backend `all_reduce(async_op=True)` makes RCCL's own stream wait for the compute stream's current position and returns, so the
    collective overlaps whatever the compute stream runs next; `finish()` reduces the buckets nobody announced and makes the compute stream
    wait for all of them (the optimiser reads the buffer next).  Bucket order of a sync-free optimisation step:
        head     F_color.6, R.*, density.beta    after the head's weight-gradient GEMMs (overlaps the colour trunk's backward)
        color_latents                            after the colour backward kernel     (overlaps the three trunk weight-gradient GEMMs)
        color_weights  F_color.0 / 2 / 4         after those GEMMs                    (overlaps the geometry latent scatter)
        geo_latents                              last (TV, pseudo-point and main-pass scatters all add into it) -> finish()
    Summation order inside a bucket is RCCL's; the result is the same sum as the single flat all-reduce."""

    def __init__(self, flat: FlatGrads, names, group=None, force=False):
        """names: parameter names in flat.params order (model.named_parameters() of the trainable tensors).
        force: reduce also on a communicator of one (see all_reduce_sum)."""
        self.flat, self.group, self.force = flat, group, bool(force)
        assert len(names) == len(flat.params)
        ranges = {}
        off = 0
        for name, p in zip(names, flat.params):
            b = self.bucket_of(name)
            lo, hi = off, off + p.numel()
            if b in ranges and ranges[b][-1][1] == lo:
                ranges[b][-1][1] = hi                     # contiguous with the bucket's previous parameter
            else:
                ranges.setdefault(b, []).append([lo, hi])
            off = hi
        self.ranges = ranges
        self._pending, self._done = [], set()
        self.log = []                                     # bucket names in the order they were reduced during the last step (tests)
        self.armed = False                                # begin() .. finish(): hooks may have started reductions that finish() must wait for
        self.timing = None                                # list -> finish() appends (event before, event after) pairs on the compute stream

    @staticmethod
    def bucket_of(name: str) -> str:
        if name == "neural_feats_color":
            return "color_latents"
        if name == "neural_feats_geometry":
            return "geo_latents"
        if name.startswith(("F_color.0.", "F_color.2.", "F_color.4.")):
            return "color_weights"
        if name.startswith(("F_color.6.", "R.")) or name == "density.beta":
            return "head"                                 # final after the head's weight-gradient GEMMs (beta: the compositing backward, earlier)
        # anything else (an unfrozen prior layer, learnable points, a new module): no hook knows when its gradient is complete, so it is
        # only ever reduced by finish(), after the whole backward
        return "rest"

    def bytes_per_step(self) -> dict:
        return {b: 4 * sum(hi - lo for lo, hi in r) for b, r in self.ranges.items()}

    def begin(self, armed=True):
        """armed=False: a backward whose gradients stay local (timing passes, graph warm-up): no hook fires, nothing for finish() to wait for."""
        self._pending, self._done, self.log = [], set(), []
        self.armed = bool(armed)

    def ready(self, name):
        if (world_size(self.group) == 1 and not self.force) or name in self._done or name not in self.ranges:
            return
        self._done.add(name)
        self.log.append(name)
        for lo, hi in self.ranges[name]:
            self._pending.append(dist.all_reduce(self.flat.buffer[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Reduce the buckets no hook announced, then make the compute stream wait for every reduction of this step."""
        ev = None
        if self.timing is not None and self.flat.buffer.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for name in self.ranges:
            self.ready(name)
        for w in self._pending:
            w.wait()
        self._pending = []
        self.armed = False
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def exposed_ms(self):
        """Per step: how long the compute stream sat in finish() (launch of the last buckets + waiting for all of them) — the part of the
        gradient exchange that did NOT hide behind the backward.  Call after a synchronisation; clears the record."""
        out = [a.elapsed_time(b) for a, b in (self.timing or [])]
        if self.timing is not None:
            self.timing = []
        return out


def fused_counts(out):
    """This rank's share of the normalisers of the mean-type losses, device float [4] = {rays, valid points, valid pseudo points, surface hits
    of the local loss}: all-reduced (16 bytes) between the forward and the fused loss kernels.  A separate function because a graphed
    multi-GPU step ends its first captured segment here (train.py)."""
    f = out["_fused"]
    dev = out["rgb_values"].device
    cnt = (f["pvalid"].bool() & f["ray_valid"].bool()).sum().float()
    if f.get("local") is not None:
        lcnt = f["local"].count()
    else:
        lcnt = out["local_count"] if "local_count" in out else torch.zeros((), device=dev)
    return torch.stack([torch.full((), float(out["rgb_values"].shape[0]), device=dev), f["n_points"][0].float(), cnt, lcnt])


def sharded_loss(loss_mod, out, ground_truth, group=None, reduce=True, force=False):
    """VolSDFLoss (spurfies/model/loss.py:51-101) on one rank's rays with GLOBAL normalisers.

    reduce=False: the counts stay this rank's own (no collective is issued) — for passes whose result is thrown away (graph warm-up /
    capture), so that a rank that re-captures on its own does not issue collectives its peers do not (round-4 advisor finding).

    Means over data-dependent counts become local sums divided by all-reduced counts:
      rgb L1 over 3R, mask BCE over R, eikonal over P, pseudo L1 over valid rendered points;
    tv depends only on replicated state, so each rank contributes tv / world."""
    dev = out["rgb_values"].device
    G = world_size(group)
    if "_fused" in out:                          # sync-free mode: fused loss kernels with the all-reduced counts as normalisers
        counts = fused_counts(out)
        if reduce:
            all_reduce_sum(counts, group, force)
        return loss_mod.fused_forward(out, ground_truth, denom=counts, world=G)
    rgb_gt = ground_truth["rgb"].to(dev).reshape(-1, 3)
    mask_gt = ground_truth["mask"].to(dev).squeeze()[:, 0][..., None]
    R_loc = out["rgb_values"].shape[0]
    g = out.get("grad_theta")
    eik_sum = ((g.norm(2, dim=1) - 1) ** 2).sum() if g is not None else torch.zeros((), device=dev)
    P_loc = torch.full((), float(0 if g is None else g.shape[0]), device=dev)
    pseudo_cnt = out.get("pseudo_count", torch.ones((), device=dev))
    lcnt = out["local_count"] if "local_count" in out else torch.zeros((), device=dev)
    counts = torch.stack([torch.full((), float(R_loc), device=dev), P_loc, pseudo_cnt.float(), lcnt])
    if reduce:
        all_reduce_sum(counts, group, force)
    R_tot, P_tot, ps_tot = counts[0], counts[1].clamp(min=1), counts[2]
    zero = torch.zeros((), device=dev)
    res = {"rgb_loss": (out["rgb_values"] - rgb_gt).abs().sum() / (3.0 * R_tot)}
    res["eikonal_loss"] = eik_sum / P_tot
    res["tv_loss"] = out["tv_loss"] / G if loss_mod.tv_weight > 0 else zero
    wsum = out["weights"].sum(-1, keepdim=True).clip(1e-3, 1.0 - 1e-3)
    res["mask_loss"] = F.binary_cross_entropy(wsum, mask_gt, reduction="sum") / R_tot
    # feature-consistency term: mean over the batch's surface hits (x sources) -> local sum / global count
    res["local_loss"] = (out["local_sum"] / counts[3].clamp(min=1.0)) if "local_sum" in out else out.get("local_loss", zero) / G
    if loss_mod.pseudo_weight > 0 and "pseudo_sum" in out:
        # no rank has a valid rendered point -> the reference's constant 1000 (split over ranks)
        res["pseudo_loss"] = torch.where(ps_tot > 0, out["pseudo_sum"] / ps_tot.clamp(min=1), torch.full_like(ps_tot, 1000.0 / G))
    else:
        res["pseudo_loss"] = zero
    res["loss"] = (loss_mod.rgb_weight * res["rgb_loss"] + loss_mod.eikonal_weight * res["eikonal_loss"]
                   + loss_mod.tv_weight * res["tv_loss"] + loss_mod.local_weight * res["local_loss"]
                   + loss_mod.pseudo_weight * res["pseudo_loss"] + res["mask_loss"])
    return res
