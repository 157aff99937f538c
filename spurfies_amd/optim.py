"""Adam over flat parameter / gradient / moment buffers, with the reference's clip and non-finite guard folded in.

The reference's step tail is `clip_grad_norm_(model.parameters(), 1.0)`, `on_after_backward()` (drop non-finite
gradients) and `torch.optim.Adam.step()` (spurfies/train.py:359-363, 548-564): ~25 foreach / reduction launches over 17
tensors.  `FlatAdam` is a `torch.optim.Adam` subclass — same constructor, param groups, `state_dict()` layout (per
parameter `step`, `exp_avg`, `exp_avg_sq`), so the reference's OptimizerParameters checkpoints load and save unchanged —
whose parameters, gradients and moments are views of four flat device buffers; `step(max_norm=...)` is then three HIP
launches (`spf_adam_step`).  With CPU parameters (host-side tests) it is torch's own clip + Adam.
"""
from __future__ import annotations

import torch

from . import _lib


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, flat_grads=None, **kw):
        """flat_grads: dist.FlatGrads over the trainable parameters (its order defines the flat layout)."""
        super().__init__(params, **kw)
        self.flat_grads = flat_grads
        self._flat = None
        if flat_grads is not None and flat_grads.params and flat_grads.params[0].is_cuda:
            self._flatten()

    # ------------------------------------------------------------------ flat views
    def _group_of(self):
        groups = [g for g in self.param_groups if len(g["params"]) > 0]
        if len(groups) != 1:
            raise ValueError("FlatAdam: exactly one non-empty param group is supported (the reference's first group is empty, train.py:157-168)")
        g = groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam: weight_decay / amsgrad / maximize are not used by the reference and not implemented")
        if [id(p) for p in g["params"]] != [id(p) for p in self.flat_grads.params]:
            raise ValueError("FlatAdam: the optimised parameters must be the FlatGrads parameters, in the same order")
        return g

    def _flatten(self):
        """Move every parameter into one flat buffer (p.data becomes a view) and create flat moment buffers whose
        per-parameter views populate `self.state` in torch.optim.Adam's layout."""
        g = self._group_of()
        ps = g["params"]
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
        dstate = torch.zeros(4, dtype=torch.float32, device=dev)                # {t, skipped, norm, clip coefficient}
        off = 0
        for p in ps:
            k = p.numel()
            flat_p[off: off + k].copy_(p.data.reshape(-1))
            st = self.state.get(p, {})
            if "exp_avg" in st:                                                 # re-flatten after load_state_dict
                m[off: off + k].copy_(st["exp_avg"].reshape(-1))
                v[off: off + k].copy_(st["exp_avg_sq"].reshape(-1))
                dstate[0] = float(st["step"])
            p.data = flat_p[off: off + k].view_as(p)
            self.state[p] = {"step": dstate[0], "exp_avg": m[off: off + k].view_as(p), "exp_avg_sq": v[off: off + k].view_as(p)}
            off += k
        ws = torch.empty(int(_lib.lib().spf_adam_workspace_floats()), dtype=torch.float32, device=dev)
        self._flat = {"param": flat_p, "m": m, "v": v, "state": dstate, "ws": ws, "n": n, "ptrs": [p.data_ptr() for p in ps]}

    def _attached(self):
        g = self._group_of()
        return self._flat is not None and [p.data_ptr() for p in g["params"]] == self._flat["ptrs"]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is not None:
            self._flatten()

    # ------------------------------------------------------------------ the step
    def step(self, closure=None, max_norm=0.0, zero_grads=False):
        """CUDA parameters: gradient-norm clip (max_norm > 0), non-finite guard and the Adam update of every parameter in
        three HIP launches (spf_adam_step), no host synchronisation (zero_grads: the sweep leaves the flat gradient buffer zero for the
        next step instead of holding the clipped gradient); returns the device tensor
        {t, skipped steps, gradient norm, clip coefficient}.  CPU parameters (host-side tests): torch's own clip + Adam."""
        if self._flat is None:
            ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if max_norm and max_norm > 0:
                torch.nn.utils.clip_grad_norm_(ps, max_norm)
            if any(not bool(torch.isfinite(p.grad).all()) for p in ps):
                return None               # train.py:548-564: "not updating model parameters"
            return super().step(closure)
        if closure is not None:
            raise NotImplementedError("FlatAdam: closures are not used by the reference")
        if not self._attached():          # parameters were re-allocated (e.g. module.to()): rebuild the flat views
            self._flatten()
        g = self._group_of()
        f = self._flat
        grad = self.flat_grads.buffer
        if grad.numel() != f["n"] or grad.device != f["param"].device:
            raise RuntimeError("FlatAdam: flat gradient buffer does not match the parameters")
        b1, b2 = g["betas"]
        with torch.cuda.device(grad.device):
            _lib.check(_lib.lib().spf_adam_step(_lib.ptr(f["param"]), _lib.ptr(grad), _lib.ptr(f["m"]), _lib.ptr(f["v"]), f["n"], float(g["lr"]),
                                                float(b1), float(b2), float(g["eps"]), float(max_norm or 0.0), 1 if zero_grads else 0, _lib.ptr(f["state"]),
                                                _lib.ptr(f["ws"]), _lib.stream_ptr()), "spf_adam_step")
        for p in g["params"]:             # the kernel wrote through raw pointers: let version-keyed caches see the change
            torch.autograd.graph.increment_version(p)
        return f["state"]
