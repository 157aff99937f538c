"""Evaluation render of fixed-size ray chunks as ONE hipGraph replay per chunk.

The evaluation forward (`PointVolSDF.forward(input, fast=-1)` under no_grad: full error-bounded sampler, kNN, SDF + normals, colour,
compositing — spurfies/model/pointneus_disent.py:614-892 as driven by train.py:414-433 / eval_spurfies.py:276-292, 864 chunks per image) has
no host synchronisation and only static shapes since the sampler's loop is controlled on the device (model/ray_sampler.py:
_z_vals_device_loop), so its ~110 launches can be captured once and replayed: the host then issues three small copies and one graph launch
per chunk instead of ~110 kernel launches.  Results are the eager forward's (same kernels, same order)."""
from __future__ import annotations

import torch


class GraphedRenderer:
    def __init__(self, model, n_rays, fast=-1):
        """model: PointVolSDF in eval mode on a CUDA device; n_rays: rays per chunk (every call must bring exactly that many)."""
        if model.training:
            raise ValueError("GraphedRenderer renders in evaluation mode: call model.eval() first")
        self.model, self.n_rays, self.fast = model, int(n_rays), fast
        self.dev = model.neural_pts.device
        self._graph = None
        self._key = None

    def _capture(self, inp):
        dev = self.dev
        self._in = {"uv": torch.zeros((1, self.n_rays, 2), dtype=torch.float32, device=dev), "pose": inp["pose"].detach().to(dev).float().clone(),
                    "intrinsics": inp["intrinsics"].detach().to(dev).float().clone()}
        self._in["uv"].copy_(inp["uv"])
        rng = torch.get_rng_state()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():          # warm-up: cell table, caches, allocator pools (the CPU generator is restored)
            for _ in range(2):
                self.model(dict(self._in, local_data=None), fast=self.fast)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.set_rng_state(rng)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._out = self.model(dict(self._in, local_data=None), fast=self.fast)
        self._flags = self.model.ray_sampler._flags
        self._key = self.model.cache_key()

    def __call__(self, inp):
        """inp: {'uv' [1,n_rays,2], 'pose' [1,4,4], 'intrinsics' [1,3|4,3|4]} -> the model's output dict (static tensors: valid until the next call)."""
        if inp["uv"].shape[1] != self.n_rays:
            raise ValueError(f"GraphedRenderer was built for chunks of {self.n_rays} rays, got {inp['uv'].shape[1]}")
        if self._graph is None or self._key != self.model.cache_key() or inp["intrinsics"].shape != self._in["intrinsics"].shape:
            self._capture(inp)
        for k in ("uv", "pose", "intrinsics"):
            self._in[k].copy_(inp[k], non_blocking=True)
        # the reference's evaluation forward consumes one torch.randint from the CPU generator per call (ray_sampler.py:562, unused by the
        # caller): drawn here, outside the graph, so the generator advances exactly as in the eager forward
        smp = self.model.ray_sampler
        torch.randint(smp.N_samples + 2 + smp.N_samples_extra, (self.n_rays,))
        self._graph.replay()
        self.model.ray_sampler._flags = self._flags             # last_iters reads this replay's flags (lazily)
        return self._out
