"""Evaluation render of fixed-size ray chunks as ONE hipGraph replay per chunk.

The evaluation forward (`PointVolSDF.forward(input, fast=-1)` under no_grad: full error-bounded sampler, kNN, SDF + normals, colour,
compositing — spurfies/model/pointneus_disent.py:614-892 as driven by train.py:414-433 / eval_spurfies.py:276-292, 864 chunks per image) has
no host synchronisation and only static shapes since the sampler's loop is controlled on the device (model/ray_sampler.py:
_z_vals_device_loop), so its ~110 launches can be captured once and replayed: the host then issues three small copies and one graph launch
per chunk instead of ~110 kernel launches.  Results are the eager forward's (same kernels, same order)."""
from __future__ import annotations

import torch

from . import ops


class GraphedRenderer:
    def __init__(self, model, n_rays, fast=-1, keys=None):
        """model: PointVolSDF in eval mode on a CUDA device; n_rays: rays per chunk (every call must bring exactly that many).
        keys: None = the reference's full output dict; a tuple out of ('rgb_values', 'depth_values', 'normal_map', 'weights') = only those
        (PointVolSDF.eval_keys: what the reference's evaluation loops read — no pseudo-point pass, TV term or plot maps in the graph)."""
        if model.training:
            raise ValueError("GraphedRenderer renders in evaluation mode: call model.eval() first")
        self.model, self.n_rays, self.fast, self.keys = model, int(n_rays), fast, (None if keys is None else tuple(keys))
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
        prev, self.model.eval_keys = self.model.eval_keys, self.keys
        try:
            with torch.cuda.stream(side), torch.no_grad():          # warm-up: cell table, caches, allocator pools (the CPU generator is restored)
                for _ in range(2):
                    self.model(dict(self._in, local_data=None), fast=self.fast)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.set_rng_state(rng)
            self._graph = torch.cuda.CUDAGraph()
            with ops.capture_guard(), torch.no_grad(), torch.cuda.graph(self._graph):
                self._out = self.model(dict(self._in, local_data=None), fast=self.fast)
        finally:
            self.model.eval_keys = prev
        torch.set_rng_state(rng)            # the captured forward drew from the CPU generator too (ray_sampler.py:562): only __call__'s draws count
        self._flags = self.model.ray_sampler._flags
        self._key = self.model.cache_key()

    def __call__(self, inp):
        """inp: {'uv' [1,n_rays,2], 'pose' [1,4,4], 'intrinsics' [1,3|4,3|4]} -> the model's output dict (static tensors: valid until the next call)."""
        if inp["uv"].shape[1] != self.n_rays:
            raise ValueError(f"GraphedRenderer was built for chunks of {self.n_rays} rays, got {inp['uv'].shape[1]}")
        if (self._graph is None or self._key != self.model.cache_key() or inp["intrinsics"].shape != self._in["intrinsics"].shape
                or inp["pose"].shape != self._in["pose"].shape):
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


class ImageRenderer:
    """A full image as a STREAM of fixed-size chunks with the outputs merged on the device — the reference's evaluation loop
    (spurfies/train.py:414-433, eval_spurfies.py:276-292: `utils.split_input` into ~864 chunks of 500 / 512 pixels, `model(s)` per chunk, a
    `.detach().cpu()` of every output per chunk, `utils.merge_output` = torch.cat over the list) without its per-chunk host work:

      * all pixel coordinates live on the device; a chunk's `uv` is gathered by row indices formed ON the device (a device-side cursor that the
        chunk itself advances), and the chunk's outputs are written straight into their rows of pre-allocated [H*W, ...] tensors
        (`index_copy_`): no list of per-chunk tensors, no clone, no cat, no host copy until the caller asks for one;
      * with graph=True cursor + gather + forward + scatter are ONE hipGraph, so the host issues one graph launch per chunk (plus the
        CPU-generator draw the reference's forward makes, to keep the generator in step);
      * a last chunk that is shorter than `n_rays` is filled up with copies of the image's last pixel, whose results land on that same pixel:
        the per-chunk convergence test of the sampler (`beta.max() > beta0`, ray_sampler.py:468) sees no new value, so every pixel gets
        exactly what the reference's shorter chunk gives it.
    Outputs: rgb_values [HW,3], depth_values [HW,1], normal_map [HW,3], weights [HW,SR] (optional)."""

    KEYS = ("rgb_values", "depth_values", "normal_map")

    def __init__(self, model, n_rays, fast=-1, graph=True, keep_weights=False):
        if model.training:
            raise ValueError("ImageRenderer renders in evaluation mode: call model.eval() first")
        self.model, self.n_rays, self.fast, self.use_graph = model, int(n_rays), fast, bool(graph)
        self.keys = self.KEYS + (("weights",) if keep_weights else ())
        self.dev = model.neural_pts.device
        self._graph, self._key, self._total = None, None, -1
        self.last_iters = []

    # ---- static state of one image size -------------------------------------------------------------------------------------------
    def _alloc(self, total, inp):
        dev, n = self.dev, self.n_rays
        self._total = total
        self._chunks = (total + n - 1) // n
        self._uv = torch.zeros((1, total, 2), dtype=torch.float32, device=dev)
        self._pose = inp["pose"].detach().to(dev).float().clone()
        self._K = inp["intrinsics"].detach().to(dev).float().clone()
        self._cursor = torch.zeros((1,), dtype=torch.int64, device=dev)
        self._lane = torch.arange(n, dtype=torch.int64, device=dev)
        self._last = torch.full((1,), total - 1, dtype=torch.int64, device=dev)
        SR = int(self.model.conf.max_shading_pts)
        widths = {"rgb_values": 3, "depth_values": 1, "normal_map": 3, "weights": SR}
        self.out = {k: torch.zeros((total, widths[k]), dtype=torch.float32, device=dev) for k in self.keys}
        self._graph = None

    def _chunk(self):
        """One chunk at the cursor: gather uv, forward, scatter, advance — every tensor op on the device, shapes static."""
        rows = torch.minimum(self._cursor + self._lane, self._last)         # a short last chunk repeats the last pixel
        uv = self._uv.index_select(1, rows)
        prev, self.model.eval_keys = self.model.eval_keys, self.keys          # only what the image needs (PointVolSDF.eval_keys)
        try:
            out = self.model({"uv": uv, "pose": self._pose, "intrinsics": self._K, "local_data": None}, fast=self.fast)
        finally:
            self.model.eval_keys = prev
        for k in self.keys:
            self.out[k].index_copy_(0, rows, out[k].reshape(self.n_rays, -1))
        self._cursor.add_(self.n_rays)

    def _capture(self):
        dev = self.dev
        rng = torch.get_rng_state()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():          # warm-up: cell table, caches, allocator pools (generator and cursor restored)
            for _ in range(2):
                self._chunk()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.set_rng_state(rng)
        self._graph = torch.cuda.CUDAGraph()
        with ops.capture_guard(), torch.no_grad(), torch.cuda.graph(self._graph):
            self._chunk()
        # the captured _chunk() ran the forward's CPU-generator draw once more (ray_sampler.py:562): restored, so that a first or re-captured
        # image advances the generator by exactly one draw per chunk, as the eager path does (round-4 advisor finding)
        torch.set_rng_state(rng)
        self._flags = self.model.ray_sampler._flags
        self._key = self.model.cache_key()

    def __call__(self, model_input, total_pixels=None, iters=False):
        """model_input: {'uv' [1,HW,2], 'pose' [1,4,4], 'intrinsics'} of ONE view -> dict of merged device tensors (valid until the next call).
        iters=True also records the realised sampler iterations per chunk in `.last_iters` (one small read-back per chunk)."""
        uv = model_input["uv"]
        total = int(total_pixels if total_pixels is not None else uv.shape[1])
        if uv.dim() != 3 or uv.shape[0] != 1 or uv.shape[1] != total:
            raise ValueError(f"ImageRenderer: uv must be [1, {total}, 2], got {tuple(uv.shape)}")
        if total != self._total or model_input["intrinsics"].shape != self._K.shape or model_input["pose"].shape != self._pose.shape:
            self._alloc(total, model_input)          # (a different pose layout, e.g. quaternion [1,7] vs [1,4,4], must not broadcast into a graph-baked buffer)
        self._uv.copy_(uv, non_blocking=True)
        self._pose.copy_(model_input["pose"], non_blocking=True)
        self._K.copy_(model_input["intrinsics"], non_blocking=True)
        graph = self.use_graph and torch.device(self.dev).type == "cuda"
        if graph and (self._graph is None or self._key != self.model.cache_key()):
            self._capture()
        self._cursor.zero_()
        smp = self.model.ray_sampler
        self.last_iters = []
        with torch.no_grad():
            for _ in range(self._chunks):
                if graph:
                    # the reference's evaluation forward consumes one torch.randint from the CPU generator per call (ray_sampler.py:562,
                    # unused by the caller): drawn here, outside the graph, so the generator advances exactly as in the eager forward
                    torch.randint(smp.N_samples + 2 + smp.N_samples_extra, (self.n_rays,))
                    self._graph.replay()
                    smp._flags = self._flags
                else:
                    self._chunk()
                if iters:
                    self.last_iters.append(smp.last_iters)
        return self.out
