"""VolSDFLoss with the reference's constructor and forward contract (spurfies/model/loss.py:19-101).

`forward(model_outputs, ground_truth)`; ground_truth = {'rgb': [1,R,3], 'mask': [1,R,3]}.
For ray-sharded multi-GPU steps `denominators` carries the GLOBAL counts so that the sum of the
ranks' losses equals the single-GPU batch loss exactly (spurfies_amd/dist.py)."""
import torch
import torch.nn.functional as F
from torch import nn

from ..utils.general import get_class


class VolSDFLoss(nn.Module):
    def __init__(self, rgb_loss, local_weight, pseudo_weight, eikonal_weight, rgb_weight=1.0, tv_weight=0.0):
        super().__init__()
        self.local_weight = local_weight
        self.pseudo_weight = pseudo_weight
        self.eikonal_weight = eikonal_weight
        self.rgb_weight = rgb_weight
        self.tv_weight = tv_weight
        self.rgb_loss = get_class(rgb_loss)(reduction="mean") if isinstance(rgb_loss, str) else rgb_loss
        self.iter_step = 0

    def get_rgb_loss(self, rgb_values, rgb_gt, model_outputs=None, t=0):
        return self.rgb_loss(rgb_values, rgb_gt.reshape(-1, 3))

    def get_eikonal_loss(self, grad_theta):
        return ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()

    def fused_forward(self, model_outputs, ground_truth, denom=None, world=1):
        """Sync-free mode: the model handed over the renderer's raw per-ray outputs (`_fused`), and every term, the weighted
        total and their gradients come from spf_loss_forward / spf_loss_backward (3 launches).  `denom` = device
        {R_total, P_total, pseudo_count_total} for ray-sharded batches (spurfies_amd/dist.py)."""
        from .. import _lib, ops

        if not (isinstance(self.rgb_loss, nn.L1Loss) and self.rgb_loss.reduction == "mean"):
            raise NotImplementedError(f"the fused loss kernels implement the reference recipe's rgb term, torch.nn.L1Loss(reduction='mean') "
                                      f"(config/ours.yaml); got {self.rgb_loss!r} — run the step with sync_free=False")
        f = model_outputs["_fused"]
        dev = model_outputs["rgb_values"].device
        rgb_gt = ground_truth["rgb"].to(dev).reshape(-1, 3).float().contiguous()
        mask = ground_truth["mask"].to(dev).float()
        R = rgb_gt.shape[0]
        mask = mask.reshape(R, -1)                      # [R,3] (the reference repeats the mask per channel) or [R,1]: column 0
        w = _lib.LossWeights(self.rgb_weight, self.eikonal_weight, self.tv_weight, self.local_weight, self.pseudo_weight, int(world))
        tv = model_outputs.get("tv_loss") if self.tv_weight > 0 else None
        psdf = f["psdf"] if self.pseudo_weight > 0 else None
        has_local = "local_sum" in model_outputs and self.local_weight > 0       # then `total` is read by an add below: no deferred finalize
        # the feature-consistency term of the step (ops.LocalTerms): summed, normalised and weighted inside the loss kernels
        total, t = ops.FusedLoss.apply(model_outputs["rgb_values"], f["acc"], psdf, tv, f["grad"], f["slot_valid"], f["n_points"],
                                       f["pvalid"], f["ray_valid"], rgb_gt, mask, mask.stride(0), w, denom, not has_local,
                                       f.get("tv_ctx") if self.tv_weight > 0 else None, f.get("local"))
        self.iter_step += 1
        local = t[5]
        if "local_sum" in model_outputs and self.local_weight > 0:
            # the feature-consistency term came through ops.LocalLoss (a step that is sync-free but whose compositing does not carry
            # ops.LocalTerms): added here, normalised by the global hit count when rays are sharded (denom[3])
            cnt = model_outputs["local_count"] if (denom is None or denom.numel() < 4) else denom[3]
            local = model_outputs["local_sum"] / cnt.clamp(min=1.0)
            total = total + self.local_weight * local
        return {"loss": total, "rgb_loss": t[1], "eikonal_loss": t[2], "tv_loss": t[3], "mask_loss": t[4], "local_loss": local,
                "pseudo_loss": t[6]}

    def forward(self, model_outputs, ground_truth):
        if "_fused" in model_outputs:
            return self.fused_forward(model_outputs, ground_truth)
        dev = model_outputs["rgb_values"].device
        rgb_gt = ground_truth["rgb"].to(dev)
        mask_gt = ground_truth["mask"].to(dev)
        zero = torch.zeros((), device=dev)
        out = {"rgb_loss": self.get_rgb_loss(model_outputs["rgb_values"], rgb_gt)}
        g = model_outputs.get("grad_theta")
        out["eikonal_loss"] = self.get_eikonal_loss(g) if g is not None else zero
        out["tv_loss"] = model_outputs["tv_loss"] if ("tv_loss" in model_outputs and self.tv_weight > 0) else zero
        if "weights" in model_outputs:
            wsum = model_outputs["weights"].sum(-1, keepdim=True)
            out["mask_loss"] = F.binary_cross_entropy(wsum.clip(1e-3, 1.0 - 1e-3), mask_gt.squeeze()[:, 0][..., None])
        else:
            out["mask_loss"] = zero
        out["local_loss"] = model_outputs.get("local_loss", zero)
        out["pseudo_loss"] = model_outputs["pseudo_pts_loss"] if ("pseudo_pts_loss" in model_outputs and self.pseudo_weight > 0) else zero
        out["loss"] = (self.rgb_weight * out["rgb_loss"] + self.eikonal_weight * out["eikonal_loss"]
                       + self.tv_weight * out["tv_loss"] + self.local_weight * out["local_loss"]
                       + self.pseudo_weight * out["pseudo_loss"] + out["mask_loss"])
        self.iter_step += 1
        return out
