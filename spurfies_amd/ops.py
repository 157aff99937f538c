"""torch-facing wrappers of the C ABI (tensor allocation + autograd glue only; every
arithmetic stage is a HIP kernel in libspurfies_hip.so)."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib, _prof

SDF_FILL = 1000.0  # pointneus_disent.py:271,371,445,703


# ---- gradient sinks (sync-free steps) -----------------------------------------------------------------
def set_grad_sinks(params, on=True):
    """Sync-free steps: the backward kernels ADD straight into the parameters' pre-zeroed .grad buffers (views of
    dist.FlatGrads' flat buffer) and hand autograd no gradient for them — no temporary gradient tensors, no zero fills,
    no AccumulateGrad add per parameter.  The caller zeroes the buffers before every backward."""
    for p in params:
        p._spf_grad_sink = p.grad if (on and p.grad is not None) else None


def _sink(p):
    g = getattr(p, "_spf_grad_sink", None)
    return g if (g is not None and g.shape == p.shape and g.is_contiguous() and g.device == p.device) else None

# ---- gradient buckets (ray-sharded steps): a callback announces that a group of gradients is final -------------------------------------
_BUCKET_HOOK = [None]


def set_bucket_hook(fn):
    """fn(name) is called by the backward code right after the launch that completes a gradient bucket ('head', 'color_latents',
    'color_weights'; spurfies_amd/dist.py:BucketedAllReduce) — only when the gradients go straight into their sinks.  None: off."""
    _BUCKET_HOOK[0] = fn


def _bucket(name):
    if _BUCKET_HOOK[0] is not None:
        _BUCKET_HOOK[0](name)


# ---- hipGraph capture: nothing may free device memory while a stream is capturing --------------------------------------------------------
class capture_guard:
    """Around every hipGraph capture: collect Python garbage BEFORE it and keep the cyclic collector off DURING it.  A dead model left over
    from an earlier run sits in reference cycles until the collector happens to run; if that is in the middle of a capture, its VoxelGrid's
    destructor calls spf_grid_destroy -> hipFree, which is illegal while any stream captures and invalidates the capture (found by
    tools/soak.py's rebuild-and-resume leg: 'operation failed due to a previous error during capture')."""

    def __enter__(self):
        import gc

        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc

        if self._was:
            gc.enable()
        return False


# ---- who owns the scratch buffers of the launches being issued ------------------------------------------------------------------------
# Workspaces (weight-gradient slabs, loss partials, fixed-point accumulators) are cached across steps.  Two steps that may run at the same
# time must not share one: eager launches are told apart by their stream, but a hipGraph is captured on torch's capture stream whatever
# stream it is later replayed on, so a stream key would bake ONE buffer into every graph of the process.  An optimisation step therefore
# names itself as the owner of everything it launches (TrainStep: `with ops.scratch_owner(self): ...`, capture and eager alike).
_SCRATCH_OWNER = [None]


_OWNER_TOKENS = [0]


def _drop_owner(token):
    """The owner died: its cached workspaces go with it (a token is never reused, unlike id(): a later step must not inherit them)."""
    for cache in (_fixed_bufs, _wgrad_ws, _loss_ws):
        for key in [k for k in cache if len(k) > 1 and isinstance(k[1], tuple) and k[1][:2] == ("owner", token)]:
            del cache[key]
    for key in [k for k in _FORK_USED if k[0] == token]:
        del _FORK_USED[key]
    for key in [k for k in _FORK_STREAMS if k[0] == token]:
        del _FORK_STREAMS[key]


class scratch_owner:
    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        token = getattr(self.owner, "_spf_scratch_token", None)
        if token is None:
            import weakref

            _OWNER_TOKENS[0] += 1
            token = self.owner._spf_scratch_token = _OWNER_TOKENS[0]
            weakref.finalize(self.owner, _drop_owner, token).atexit = False      # at interpreter exit the caches go with the process
        self.prev, _SCRATCH_OWNER[0] = _SCRATCH_OWNER[0], token
        return self

    def __exit__(self, *exc):
        _SCRATCH_OWNER[0] = self.prev
        return False


def _scratch_key(dev):
    """(device, owner[, branch]) of the scratch buffers of the launch being issued: the naming optimisation step (and the forked branch the
    launch is issued on: two branches of one step may run at the same time), else the current stream."""
    own = _SCRATCH_OWNER[0]
    if own is None:
        return (dev.index, ("stream", torch.cuda.current_stream(dev).cuda_stream))
    return (dev.index, ("owner", own, _BRANCH[0]))


# ---- forked branches of one step -------------------------------------------------------------------------------------------------------
# Passes of the optimisation step that do not depend on each other are issued on side streams: captured into the step's hipGraph they become
# parallel branches, so that the latency-bound chains of a SMALL per-GPU batch (strong scaling: 128 rays per rank) overlap instead of queueing
# behind each other — the pseudo-point pass (kNN + compaction + geometry kernel on 128 points: one tile pass on 16 CUs) and the TV term beside
# the per-point head (102 of 256 CUs), the head's weight-gradient GEMMs beside the colour backward, the latent scatters of the geometry passes
# beside the colour trunk.  Off by default (TrainStep(fork=...) turns it on for its own launches); with it off every `branch` is a no-op
# and the launches stay on the caller's stream in program order.
_FORK = [False]
_BRANCH = [None]
_FORK_STREAMS = {}
_FORK_USED = {}
_KEEP = []


def set_fork(on=True):
    prev, _FORK[0] = _FORK[0], bool(on)
    return prev


def fork_enabled() -> bool:
    """(the bit-reproducible mode keeps one stream: its fixed-point accumulators are flushed in program order)"""
    return _FORK[0] and _SCATTER["mode"] != "fixed"


def keep(*tensors):
    """Hold tensors that a launch on ANOTHER stream than their allocation's reads until the step's branches have been joined: the caching
    allocator hands a freed block to the next allocation on its own stream straight away (also during hipGraph capture), which may run beside
    the side-stream reader."""
    if fork_enabled():
        _KEEP.append(tensors)


class branch:
    """`with ops.branch(name, device) as b:` — the enclosed launches go to the side stream `name` of the current step (it first waits for the
    caller's stream).  b.record() -> an event the caller's stream can wait for (`ops.wait(ev)`) without waiting for the rest of the
    branch; ops.join(device) makes the caller's stream wait for every branch used since the last join."""

    def __init__(self, name, dev, after=None):
        """after: an event of the caller's stream (ops.mark) the branch starts behind, instead of the caller's current position — the caller
        can then issue its OWN continuation first and the branch afterwards: in a captured graph the node issued first stays on the parent's
        queue and later children get queues of their own (ROCm's graph executor, measured: profiles/r05_fork_*.txt)."""
        self.name, self.dev, self.after = name, torch.device(dev), after
        self.on = fork_enabled() and self.dev.type == "cuda" and _BRANCH[0] is None        # no nested forks
        self.stream = None

    def __enter__(self):
        if not self.on:
            return self
        key = (_SCRATCH_OWNER[0], self.dev.index, self.name)
        st = _FORK_STREAMS.get(key)
        if st is None:
            st = _FORK_STREAMS[key] = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        if self.after is not None:
            st.wait_event(self.after)
        else:
            st.wait_stream(main)
        self.stream = st
        _FORK_USED.setdefault((_SCRATCH_OWNER[0], self.dev.index), {})[self.name] = st
        self._prev_branch, _BRANCH[0] = _BRANCH[0], self.name
        self._ctx = torch.cuda.stream(st)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
            _BRANCH[0] = self._prev_branch
        return False

    def record(self):
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev


def wait(ev):
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def mark(dev):
    """An event at the caller's current position (None when forking is off): `branch(..., after=ev)` starts there."""
    if not (fork_enabled() and torch.device(dev).type == "cuda"):
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(torch.device(dev)))
    return ev


def join(dev):
    """The caller's stream waits for every branch of the current step used since the last join (no-op when none was)."""
    dev = torch.device(dev)
    used = _FORK_USED.pop((_SCRATCH_OWNER[0], dev.index), None)
    if used:
        main = torch.cuda.current_stream(dev)
        for st in used.values():
            main.wait_stream(st)
    _KEEP.clear()


# ---- optional per-launch timing (bench.py roofline): spurfies_amd/_prof.py -----------------------
def profile_start(tags=None):
    _prof.start(tags)


def profile_stop():
    """-> [{'tag', 'ms', ...}] per profiled launch since profile_start(): 'geo' (pairs, rows, with_grad), 'knn' (rays, samples_per_ray,
    slots, hits), 'render_fwd' / 'render_bwd' (rays, slots), 'color_fwd' / 'color_bwd' (pairs)."""
    return _prof.stop()



def compact_points(slot_valid, fill_sdf=None, fill_grad=None):
    """slot_valid uint8 [R,SR] -> (point_slot i32 [R*SR], slot_point i32 [R*SR], n_points i32 [1]); no host sync.
    fill_sdf [R*SR] / fill_grad [R*SR,3] (optional, uninitialised): set to the 1000 filler / 0 on the way, so that geo_forward's
    output buffers need no separate fill launches."""
    R, SR = slot_valid.shape
    dev = slot_valid.device
    point_slot = torch.empty((R * SR,), dtype=torch.int32, device=dev)
    slot_point = torch.empty((R * SR,), dtype=torch.int32, device=dev)
    n_points = torch.empty((2,), dtype=torch.int32, device=dev)[:1]     # first half of a [n_points, n_pairs] pair: PairList adopts it
    scratch = torch.empty((R + 1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_compact_points(_lib.ptr(slot_valid), R, SR, _lib.ptr(point_slot), _lib.ptr(slot_point),
                                                 _lib.ptr(n_points), _lib.ptr(scratch), _lib.ptr(fill_sdf), SDF_FILL, _lib.ptr(fill_grad),
                                                 _lib.stream_ptr()), "spf_compact_points")
    return point_slot, slot_point, n_points


_COMPACT_SYNC = {}
_COMPACT_ONE_LAUNCH = [True]


def set_compact_one_launch(on=True):
    """spf_compact_pairs as one launch (chunks publish their totals to each other; default) or as the count + write pair (tests compare both)."""
    _COMPACT_ONE_LAUNCH[0] = bool(on)


def compact_sync_words(n_slots) -> int:
    """Words of the zero-on-entry / zero-on-exit buffer spf_compact_pairs' one-launch form needs for `n_slots` slots."""
    return max(int(_lib.lib().spf_compact_sync_words(int(n_slots))), 2049)


class CompactSync:
    """OWNED word buffers of the one-launch compaction (include/spurfies_hip.h: spf_compact_pairs, `sync`): one per (owner, role).  A buffer
    must never be shared by two compaction launches that may run at the same time — the launch spins on words its own blocks publish and the
    last block clears them, so a second launch on the same words hangs or mixes the lists.  A stream-keyed global cannot promise that once
    launches are captured into hipGraphs (every capture sees torch's capture stream, the replays run wherever they are launched: round-4
    advisor finding), so the model owns one buffer per kNN pass ('sampler', 'main', 'points', ...) and hands it to PairList.from_slots."""

    def __init__(self):
        self._bufs = {}

    def get(self, role, dev, n_slots):
        if not _COMPACT_ONE_LAUNCH[0]:
            return None
        need = compact_sync_words(n_slots)
        key = (role, torch.device(dev).index)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = self._bufs[key] = torch.zeros((need,), dtype=torch.int64, device=dev)
        return buf


def _compact_sync(dev, n_slots):
    """Fallback for callers that own no CompactSync (tests, one-off queries): one buffer per (device, stream), eager launches only — under
    hipGraph capture the stream is torch's capture stream whatever stream the replay will run on, so the two-launch form (sync = NULL)
    is used there instead of a buffer some other graph may share."""
    if not _COMPACT_ONE_LAUNCH[0] or torch.cuda.is_current_stream_capturing():
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    need = compact_sync_words(n_slots)
    buf = _COMPACT_SYNC.get(key)
    if buf is None or buf.numel() < need:
        buf = _COMPACT_SYNC[key] = torch.zeros((need,), dtype=torch.int64, device=dev)
    return buf


class PairList:
    """Device-side lists of the valid points and of their (point, neighbour) pairs — the rows of the MLP kernels
    (utils.py:96-113, 172-183 without the masked_select host syncs).  `counts` = [n_points, n_pairs] on the device."""

    def __init__(self, nbr, point_slot, n_points):
        rows, k = nbr.shape
        dev = nbr.device
        self.nbr, self.point_slot, self.k = nbr, point_slot, k
        self.max_points = rows if point_slot is None else min(rows, point_slot.shape[0])
        self.max_pairs = self.max_points * k
        base = getattr(n_points, "_base", None) if n_points is not None else None
        if base is not None and base.dtype == torch.int32 and base.numel() == 2 and n_points.data_ptr() == base.data_ptr():
            self.counts = base                  # compact_points() left room for n_pairs next to its count: no copy launch
        else:
            self.counts = torch.empty((2,), dtype=torch.int32, device=dev)
            if n_points is None:
                self.counts[0] = self.max_points
            else:
                self.counts[:1].copy_(n_points)
        self.n_points, self.n_pairs = self.counts[:1], self.counts[1:]
        self.pair_off = torch.empty((self.max_points + 1,), dtype=torch.int32, device=dev)
        self.pair_point = torch.empty((self.max_pairs,), dtype=torch.int32, device=dev)
        scratch = torch.empty((self.max_points // 2048 + 2,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_build_pairs(_lib.ptr(nbr), _lib.ptr(point_slot), _lib.ptr(self.n_points), self.max_points, k,
                                                  _lib.ptr(self.pair_off), _lib.ptr(self.pair_point), _lib.ptr(self.n_pairs),
                                                  _lib.ptr(scratch), _lib.stream_ptr()), "spf_build_pairs")

    @classmethod
    def from_slots(cls, slot_valid, nbr, fill_sdf=None, fill_grad=None, gate=None, sync=None, filt=None):
        """Valid-point compaction AND the pair list of a kNN result in one pair of launches (spf_compact_pairs): slot_valid uint8 [R,SR],
        nbr int32 [R*SR,k] indexed by slot; optional uninitialised fill_sdf [R*SR] / fill_grad [R*SR,3] receive the 1000 filler / zeros.
        gate: device int32 [1]; 0 reports empty lists (the MLP kernels behind this pass then do nothing).
        sync: the caller's OWN zero-on-entry word buffer for the one-launch form (CompactSync.get); None = a per-stream fallback buffer
        (eager launches) or the two-launch form (under graph capture).
        filt: (loc [R,SR,3], cam_loc [R,3], ray_dirs [R,3]) — filter_points (pointneus_disent.py:207-239) rides in the same launch;
        its outputs land in self.filtered = (z [R,SR], deltas [R,SR], x [R*SR,3])."""
        R, SR = slot_valid.shape
        rows, k = nbr.shape
        dev = nbr.device
        self = cls.__new__(cls)
        self.nbr, self.k = nbr, k
        self.point_slot = torch.empty((rows,), dtype=torch.int32, device=dev)
        self.slot_point = torch.empty((rows,), dtype=torch.int32, device=dev)
        self.max_points, self.max_pairs = rows, rows * k
        self.counts = torch.empty((2,), dtype=torch.int32, device=dev)
        self.n_points, self.n_pairs = self.counts[:1], self.counts[1:]
        self.pair_off = torch.empty((rows + 1,), dtype=torch.int32, device=dev)
        self.pair_point = torch.empty((self.max_pairs,), dtype=torch.int32, device=dev)
        scratch = torch.empty((2 * (rows // 2048 + 2),), dtype=torch.int32, device=dev)
        sync_buf = sync if (sync is not None and _COMPACT_ONE_LAUNCH[0]) else _compact_sync(dev, rows)
        self.filtered = None
        with torch.cuda.device(dev):
            if filt is not None:
                loc, cam_loc, ray_dirs = (t.detach().contiguous() for t in filt)
                z = torch.empty((R, SR), dtype=torch.float32, device=dev)
                deltas = torch.empty((R, SR), dtype=torch.float32, device=dev)
                xs = torch.empty((R * SR, 3), dtype=torch.float32, device=dev)
                _lib.check(_lib.lib().spf_compact_pairs_filter(_lib.ptr(slot_valid), _lib.ptr(nbr), R, SR, k, _lib.ptr(self.point_slot), _lib.ptr(self.slot_point),
                                                               _lib.ptr(self.pair_off), _lib.ptr(self.pair_point), _lib.ptr(self.counts), _lib.ptr(scratch),
                                                               _lib.ptr(fill_sdf), SDF_FILL, _lib.ptr(fill_grad), _lib.ptr(gate), _lib.ptr(sync_buf),
                                                               _lib.ptr(loc), _lib.ptr(cam_loc), _lib.ptr(ray_dirs), _lib.ptr(z), _lib.ptr(deltas), _lib.ptr(xs),
                                                               _lib.stream_ptr()), "spf_compact_pairs_filter")
                self.filtered = (z, deltas, xs)
            else:
                _lib.check(_lib.lib().spf_compact_pairs(_lib.ptr(slot_valid), _lib.ptr(nbr), R, SR, k, _lib.ptr(self.point_slot), _lib.ptr(self.slot_point),
                                                        _lib.ptr(self.pair_off), _lib.ptr(self.pair_point), _lib.ptr(self.counts), _lib.ptr(scratch),
                                                        _lib.ptr(fill_sdf), SDF_FILL, _lib.ptr(fill_grad), _lib.ptr(gate), _lib.ptr(sync_buf),
                                                        _lib.stream_ptr()), "spf_compact_pairs")
        return self

    def host_counts(self):
        """(P, n_pairs) on the host — ONE synchronising copy."""
        c = self.counts.tolist()
        return int(c[0]), int(c[1])


# Arithmetic of the MLP / weight-gradient kernels (include/spurfies_hip.h: SPF_ARITH_*), chosen per call; this table is host-side
# state of the Python layer only — the C ABI itself keeps none.  'split' (default): fp32-class products (<= 2 ulp per product) from three bf16 pieces per
# operand on the bf16 matrix pipe; 'f32': fp32 MFMA (the verification twin).
_ARITH = {"geo": 3, "color": 0, "rhead": 0, "wgrad": 0}        # geometry: H2 (round 6); the others: the 'split' family, H2 inside it per kernel (_H2)
_ARITH_NAMES = {"split": 0, "f32": 1}
# H2 arithmetic (round 6; include/spurfies_hip.h: SPF_ARITH_H2) inside the 'split' family of the colour / head / weight-gradient kernels: per kernel,
# the piece products are three fp16 ones (two fp16 pieces per operand, main + cross accumulators) instead of six bf16 ones; everything else of the
# 'split' family — operand layouts, sign words, who forms the bias gradients — is unchanged, so forward and backward may differ in it.
_H2 = {"color_fwd": True, "color_bwd": True, "rhead_fwd": True, "rhead_bwd": True, "wgrad": True}


if os.environ.get("SPF_H2_FLAGS"):          # same-box A/B runs of whole programs: SPF_H2_FLAGS='{"color_bwd": false}'
    import json as _json

    _H2.update({k: bool(v) for k, v in _json.loads(os.environ["SPF_H2_FLAGS"]).items() if k in _H2})


def set_h2(**flags):
    """ops.set_h2(color_fwd=False, ...): switch single kernels of the 'split' family between H2 (three fp16 piece products) and the six bf16 piece
    products (tests, same-box A/B runs).  -> the previous flags."""
    prev = dict(_H2)
    for k, v in flags.items():
        if k not in _H2:
            raise KeyError(k)
        _H2[k] = bool(v)
    return prev


def _arith_of(family_arith, kernel):
    return 3 if (family_arith == 0 and _H2[kernel]) else family_arith
_GEO_ARITH_NAMES = {"split": 0, "f32": 1, "split_w": 2, "h2": 3}     # 'split_w': the split arithmetic on 32x32x16 MFMA tiles; 'h2': two fp16 pieces (SPF_ARITH_H2)


def set_geo_mode(mode: str):
    """'h2' (default since round 6: two fp16 pieces per operand, three exact piece products per fp32 product on v_mfma_f32_32x32x16_f16 — the
    same measured error against float64 as the fp32-MFMA kernel at half the matrix instructions of the bf16 scheme, tools/engine_accuracy.py),
    'split' (six bf16-piece products on v_mfma_f32_16x16x32_bf16), 'split_w' (the same products on v_mfma_f32_32x32x16_bf16
    tiles — bench.py times both on the box it runs on) or 'f32' (fp32 MFMA, verification twin)."""
    _ARITH["geo"] = _GEO_ARITH_NAMES[mode]


_GEO_CLOCK = [False]


def geo_clock_enable(on=True):
    """Ask the bf16-piece geometry launches of THIS process to stamp the library's held-clock counters (arith | SPF_ARITH_CLOCK).  Off by
    default: the counters are process-wide per device (include/spurfies_hip.h), so only a single measuring caller (bench.py) turns them on."""
    _GEO_CLOCK[0] = bool(on)


def geo_clock(reset=True):
    """Shader clock the bf16-piece geometry kernels HELD since the last reset, measured by the kernels themselves
    (include/spurfies_hip.h: spf_geo_clock_read; synchronises the device) ->
    {('split' | 'split_w', with_jacobian): {'ghz', 'workgroups', 'mean_us_per_workgroup'}} for the combinations that ran."""
    import ctypes

    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 12)()
    _lib.check(_lib.lib().spf_geo_clock_read(ctypes.cast(buf, ctypes.c_void_p), 1 if reset else 0), "spf_geo_clock_read")
    out = {}
    for e, name in enumerate(("split", "split_w")):
        for j in (0, 1):
            cyc, ticks, wgs = (int(buf[(e * 2 + j) * 3 + i]) for i in range(3))
            if wgs:
                out[(name, bool(j))] = {"ghz": 0.1 * cyc / max(ticks, 1), "workgroups": wgs, "mean_us_per_workgroup": ticks / wgs / 100.0}
    return out


def geo_mode() -> str:
    return {v: k for k, v in _GEO_ARITH_NAMES.items()}[_ARITH["geo"]]


def pack_geometry_weights(state: dict) -> torch.Tensor:
    """state: {'F_geometry.0.weight', ..., 'T.0.bias'} CUDA float32 tensors -> packed image."""
    names = ["F_geometry.0", "F_geometry.2", "F_geometry.4", "F_geometry.6", "F_geometry.8", "T.0"]
    args = []
    for n in names:
        w, b = state[n + ".weight"], state[n + ".bias"]
        args += [w.detach().contiguous().float(), b.detach().contiguous().float()]
    dev = args[0].device
    packed = torch.empty((int(_lib.lib().spf_geo_packed_floats()),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_geo_pack(*[_lib.ptr(a) for a in args], _lib.ptr(packed), _lib.stream_ptr()), "spf_geo_pack")
    return packed


def geo_forward(x, pl: "PairList", pts, feat_geo, packed, rbf, with_grad, sdf_out=None, grad_out=None, reduce=True, lite=False):
    """Rows = first dim of x / nbr.  Returns dict(sdf [rows] (1000 where not a valid point), grad [rows,3] | None,
    wn [max_pairs], jac [max_pairs,32] | None).  sdf_out / grad_out: buffers already holding the filler (compact_points(fill_*)).
    reduce=False (no grad): the per-point reduction is left to the consumer (sampler_train); the result is 'pair_tmp' [max_pairs,5].
    lite (no grad, 'split_w' engine only; ignored otherwise): reduced products, SPF_ARITH_LITE — for sampler passes, opt-in."""
    rows = pl.nbr.shape[0]
    dev = x.device
    if not reduce:
        if with_grad:
            raise ValueError("geo_forward(reduce=False) is the SDF-only form")
        sdf = None
    else:
        sdf = sdf_out if sdf_out is not None else torch.full((rows,), SDF_FILL, dtype=torch.float32, device=dev)
    wn = torch.empty((pl.max_pairs,), dtype=torch.float32, device=dev) if reduce else None
    grad = (grad_out if grad_out is not None else torch.zeros((rows, 3), dtype=torch.float32, device=dev)) if with_grad else None
    jac = torch.empty((pl.max_pairs, 32), dtype=torch.float32, device=dev) if with_grad else None
    tmp = torch.empty((pl.max_pairs, 5), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _prof.span("geo", pairs=pl.n_pairs, rows=rows, with_grad=bool(with_grad)):
        _lib.check(_lib.lib().spf_geo_forward(_lib.ptr(x), _lib.ptr(pl.nbr), _lib.ptr(pl.point_slot), _lib.ptr(pl.pair_off), _lib.ptr(pl.pair_point),
                                              _lib.ptr(pl.n_points), _lib.ptr(pl.n_pairs), pl.max_points, pl.max_pairs, pl.k, _lib.ptr(pts),
                                              _lib.ptr(feat_geo), _lib.ptr(packed), float(rbf), _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(wn),
                                              _lib.ptr(jac), _lib.ptr(tmp),
                                              _ARITH["geo"] | (0x100 if _GEO_CLOCK[0] else 0) | (0x200 if (lite and not with_grad and _ARITH["geo"] == 2) else 0),
                                              _lib.stream_ptr()), "spf_geo_forward")
    return {"sdf": sdf, "wn": wn, "grad": grad, "jac": jac, "pair_tmp": tmp}


# ---- latent-gradient scatter: float atomics (default) or order-independent fixed-point accumulation --------------------------
_SCATTER = {"mode": "atomic"}
_fixed_bufs = {}


def set_scatter_mode(mode: str):
    """'atomic' (default): float atomics, sums depend on the order the atomics land in (last-bit run-to-run noise, like the
    reference's index_add_).  'fixed': every term is added as a 2^-48 fixed-point integer with 64-bit integer atomics and the sum is
    rounded to fp32 once (include/spurfies_hip.h: spf_fixed_accumulate) — latent gradients are then bit-reproducible run to run."""
    if mode not in ("atomic", "fixed"):
        raise ValueError(mode)
    _SCATTER["mode"] = mode


def scatter_mode() -> str:
    return _SCATTER["mode"]


def _fixed_acc(like, role):
    """Zero int64 accumulator shaped like the gradient buffer `like`: a view of ONE flat buffer per (device, stream, role) that grows to the
    largest size asked for (exact-size buffers change shape from step to step in the default mode: a buffer per shape would grow without bound).
    spf_fixed_accumulate leaves every entry it flushed zero again, so any prefix is ready for the next use; two accumulators that are alive at
    the same time must have different roles."""
    key = _scratch_key(like.device) + (role,)
    n = like.numel()
    buf = _fixed_bufs.get(key)
    if buf is None or buf.numel() < n + 1:
        # one int64 more IN FRONT of the accumulators: the buffer's status word (acc[-1], include/spurfies_hip.h: buffer contract)
        buf = _fixed_bufs[key] = torch.zeros((n + 1,), dtype=torch.int64, device=like.device)
    return buf[1: n + 1].view(like.shape)


def _fixed_flush(acc, dst):
    with torch.cuda.device(dst.device):
        _lib.check(_lib.lib().spf_fixed_accumulate(_lib.ptr(acc), _lib.ptr(dst), dst.numel(), _lib.stream_ptr()), "spf_fixed_accumulate")


def geo_backward_latents(g_sdf, wn, jac, pl: "PairList", g_feat_geo, grad_x=None, g_x=None):
    """grad_x [rows,3] (the forward's d sdf / d x) + g_x [rows,3] (uninitialised): g_x = g_sdf[:, None] * grad_x is formed by the same launch."""
    acc = _fixed_acc(g_feat_geo, "geo_latents") if _SCATTER["mode"] == "fixed" else None
    with torch.cuda.device(g_sdf.device):
        _lib.check(_lib.lib().spf_geo_backward_latents(_lib.ptr(g_sdf), _lib.ptr(wn), _lib.ptr(jac), _lib.ptr(pl.nbr), _lib.ptr(pl.point_slot),
                                                       _lib.ptr(pl.pair_off), _lib.ptr(pl.pair_point), _lib.ptr(pl.n_pairs), pl.max_pairs, pl.k,
                                                       _lib.ptr(g_feat_geo), _lib.ptr(acc), _lib.ptr(grad_x), _lib.ptr(g_x),
                                                       0 if g_x is None else g_x.shape[0], _lib.stream_ptr()), "spf_geo_backward_latents")
    if acc is not None:
        _fixed_flush(acc, g_feat_geo)
    return g_feat_geo


class GeoSDF(torch.autograd.Function):
    """(sdf, d sdf/d x, normalised RBF weights per pair) through the fused kernel.  sdf is differentiable
    w.r.t. x (d sdf/d x is the kernel's `grad` output — RBF weights are detached in the reference,
    pointneus_disent.py:242) and w.r.t. the geometry latent table (scalar-output MLP:
    d sdf_j/d latent_j is the Jacobian row the forward sweep stored)."""

    @staticmethod
    def forward(ctx, x, feat_geo, pl, pts, packed, rbf, sdf_out=None, grad_out=None):
        if sdf_out is not None:          # caller-provided output buffers (already holding the filler) are written in place and returned
            ctx.mark_dirty(sdf_out, grad_out)
        res = geo_forward(x.detach(), pl, pts, feat_geo.detach(), packed, rbf, with_grad=True, sdf_out=sdf_out, grad_out=grad_out)
        ctx.save_for_backward(res["wn"], res["jac"], res["grad"])
        ctx.pl = pl
        ctx.n_table = feat_geo.shape[0]
        ctx.sink = _sink(feat_geo)
        ctx.mark_non_differentiable(res["grad"], res["wn"])
        ctx.set_materialize_grads(False)      # no zero tensors (two fill launches per pass) for the two non-differentiable outputs
        return res["sdf"], res["grad"], res["wn"]

    @staticmethod
    def backward(ctx, g_sdf, _g_grad, _g_wn):
        wn, jac, grad = ctx.saved_tensors
        if g_sdf is None:
            return (None,) * 8
        g_sdf = g_sdf.contiguous()
        g_x = g_feat = None
        want_x = ctx.needs_input_grad[0]
        if ctx.needs_input_grad[1]:
            if want_x:                         # rides along in the latent-gradient launch
                g_x = torch.empty_like(grad)
            if ctx.sink is not None and not want_x and _BRANCH[0] is None:
                # a forked step: the main pass's latent scatter (nothing on this stream reads its result before the optimiser) on a branch
                with branch("geo_scatter", g_sdf.device):
                    geo_backward_latents(g_sdf, wn, jac, ctx.pl, ctx.sink, None, None)
                    keep(g_sdf, wn, jac, ctx.pl)
            elif ctx.sink is not None:
                geo_backward_latents(g_sdf, wn, jac, ctx.pl, ctx.sink, grad if want_x else None, g_x)
            else:
                g_feat = torch.zeros((ctx.n_table, 32), dtype=torch.float32, device=g_sdf.device)
                geo_backward_latents(g_sdf, wn, jac, ctx.pl, g_feat, grad if want_x else None, g_x)
        elif want_x:
            g_x = g_sdf.unsqueeze(-1) * grad
        return g_x, g_feat, None, None, None, None, None, None


class GatherRows(torch.autograd.Function):
    """table[idx] for a [N,C] latent table (C in {32, 64}); backward = HIP row scatter-add
    (the reference's index_select backward, spurfies/model/utils.py:158-161)."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return table.detach()[idx.long()]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, c = ctx.shape
        g = g.contiguous().view(-1, c)
        out = torch.zeros((n, c), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().spf_scatter_add_rows(_lib.ptr(g), _lib.ptr(idx), g.shape[0], c, _lib.ptr(out), _lib.stream_ptr()),
                       "spf_scatter_add_rows")
        return out, None


def gather_rows(table, idx_i32):
    """idx_i32: int32 tensor of any shape (all >= 0) -> table rows, shape idx.shape + [C]."""
    flat = idx_i32.reshape(-1).contiguous()
    return GatherRows.apply(table, flat).view(*idx_i32.shape, table.shape[1])


class TVLoss(torch.autograd.Function):
    """mean_i tv_i over the static neighbour graph (spurfies/model/utils.py:221-282), HIP fwd + bwd."""

    @staticmethod
    def forward(ctx, feat, nbr, w, norm, reduce=True):
        """reduce=True: mean_i tv_i (utils.py:282).  reduce=False: the per-point terms tv [n] — for FusedLoss, whose kernels form the mean
        themselves (one reduction launch less) and hand back ONE gradient value for all points (an expanded, stride-0 tensor)."""
        n, k = nbr.shape
        feat_c = feat.detach().contiguous()
        tv = torch.empty((n,), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().spf_tv_forward(_lib.ptr(feat_c), _lib.ptr(nbr), _lib.ptr(w), _lib.ptr(norm), n, k, _lib.ptr(tv),
                                                 _lib.stream_ptr()), "spf_tv_forward")
        ctx.save_for_backward(feat_c, nbr, w, norm)
        ctx.sink = _sink(feat)
        ctx.reduce = reduce
        return tv.mean() if reduce else tv

    @staticmethod
    def backward(ctx, g):
        feat, nbr, w, norm = ctx.saved_tensors
        n, k = nbr.shape
        if ctx.reduce:
            g_tv, stride, scale = g.detach().reshape(1).contiguous(), 0, 1.0 / n      # one device scalar: d mean / d tv_i = g / n for every point
        elif g.stride(0) == 0:
            g_tv, stride, scale = g.detach()[:1].contiguous(), 0, 1.0                    # FusedLoss: the same d loss / d tv_i for every point
        else:
            g_tv, stride, scale = g.detach().contiguous(), 1, 1.0
        out = ctx.sink if ctx.sink is not None else torch.zeros_like(feat)
        acc = _fixed_acc(out, "geo_latents") if _SCATTER["mode"] == "fixed" else None
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().spf_tv_backward(_lib.ptr(feat), _lib.ptr(nbr), _lib.ptr(w), _lib.ptr(norm), _lib.ptr(g_tv), stride, scale, n, k,
                                                  _lib.ptr(out), _lib.ptr(acc), _lib.stream_ptr()), "spf_tv_backward")
        if acc is not None:
            _fixed_flush(acc, out)
        return (None if ctx.sink is not None else out), None, None, None, None


# ---- fused colour-feature path ------------------------------------------------------------------
_C_ORIG = None


def _color_col_perm(device):
    """internal column k of the colour kernels -> reference column (spf_color_pack's c_orig)."""
    global _C_ORIG
    if _C_ORIG is None or _C_ORIG.device != device:
        k = torch.arange(103, device=device)
        _C_ORIG = torch.where(k < 64, 39 + k, k - 64)
    return _C_ORIG


def prepack_color(fc, n_points, dev):
    """The colour trunk's weight image + the zeroed [n_points,256] accumulator of the RBF-weighted mean, formed AHEAD of the forward's
    colour stage (a forked branch packs while the geometry kernel runs) -> `pre` of ColorAgg.apply."""
    agg3 = torch.empty((n_points, 256), dtype=torch.float32, device=dev)
    return pack_color_weights([fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias], zero=agg3), agg3


def prepack_rhead(fc6, rh, n_rows, dev):
    """The head stage's weight image + the zeroed dense colour array [n_rows,3] -> `pre` of RHead.apply."""
    colors = torch.empty((n_rows, 3), dtype=torch.float32, device=dev)
    return pack_rhead_weights([fc6.weight, fc6.bias, rh[0].weight, rh[0].bias, rh[2].weight, rh[2].bias, rh[4].weight, rh[4].bias], zero=colors), colors


def pack_color_weights(ws, zero=None):
    """ws: [w0, b0, w2, b2, w4, b4] of F_color's three activated layers -> packed image.  zero: a float32 buffer cleared by the same launch."""
    args = [t.detach().contiguous().float() for t in ws]
    dev = args[0].device
    packed = torch.empty((int(_lib.lib().spf_color_packed_floats()),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_color_pack(*[_lib.ptr(a) for a in args], _lib.ptr(packed), _lib.ptr(zero), 0 if zero is None else zero.numel(),
                                             _lib.stream_ptr()), "spf_color_pack")
    return packed


class _GradModeFunction(torch.autograd.Function):
    """autograd.Function whose forward needs to know whether a backward can follow: `ctx.needs_input_grad` only mirrors the
    inputs' requires_grad (also under torch.no_grad()), and grad mode is always off inside forward()."""

    _outer_grad_mode = True

    @classmethod
    def apply(cls, *args, **kwargs):
        prev, _GradModeFunction._outer_grad_mode = _GradModeFunction._outer_grad_mode, torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _GradModeFunction._outer_grad_mode = prev


class ColorAgg(_GradModeFunction):
    """agg3[p,256] = sum_j wn_j a3_j for the P valid points, a3 = the third activation of F_color on
    [posenc6(x_pi) | colour latent] (pointneus_disent.py:325-336).  F_color's last layer is linear and commutes with the
    weighted mean, so it runs once per point inside `RHead`.  Forward and the data-gradient chain are HIP kernels; the
    weight gradients are spf_wgrad launches over the activation / pre-activation-gradient buffers the kernels store."""

    @staticmethod
    def forward(ctx, feat_col, w0, b0, w2, b2, w4, b4, x, wn, pl, pts, n_valid, n_pairs, pre=None):
        """n_valid / n_pairs: host ints (buffers sized exactly), or None = sync-free mode: worst-case buffers, every
        kernel (incl. the wgrad GEMMs) reads the counts from device memory.  pre: (packed image, zeroed agg3 [max_points,256]) from
        prepack_color (sync-free mode only)."""
        dev = x.device
        ctx.static = n_valid is None
        P, NP = (pl.max_points, pl.max_pairs) if ctx.static else (int(n_valid), int(n_pairs))
        tiles = (NP + 63) // 64
        rows = 64 * tiles
        if pre is not None and ctx.static and pre[1].shape[0] == P:
            packed, agg3 = pre
        else:
            agg3 = torch.empty((P, 256), dtype=torch.float32, device=dev)          # cleared by the packing launch
            packed = pack_color_weights([w0, b0, w2, b2, w4, b4], zero=agg3)
        train = _GradModeFunction._outer_grad_mode and any(ctx.needs_input_grad[:7])     # no training stores under torch.no_grad()
        if train:
            bufs = [torch.empty((rows, 104), dtype=torch.float32, device=dev), torch.empty((rows, 256), dtype=torch.float32, device=dev),
                    torch.empty((rows, 256), dtype=torch.float32, device=dev), torch.empty((tiles, 3, 512), dtype=torch.int32, device=dev)]
        else:
            bufs = [None] * 4
        acc = _fixed_acc(agg3, "agg3") if (_SCATTER["mode"] == "fixed" and _ARITH["color"] == 0) else None
        with torch.cuda.device(dev), _prof.span("color_fwd", pairs=pl.n_pairs, train=bool(train)):
            _lib.check(_lib.lib().spf_color_forward(_lib.ptr(x), _lib.ptr(pl.nbr), _lib.ptr(wn), _lib.ptr(pl.point_slot), _lib.ptr(pl.pair_off),
                                                    _lib.ptr(pl.pair_point), _lib.ptr(pl.n_pairs), NP, pl.k, _lib.ptr(pts),
                                                    _lib.ptr(feat_col.detach()), _lib.ptr(packed), _lib.ptr(agg3),
                                                    *[_lib.ptr(a) for a in bufs], _lib.ptr(acc), _arith_of(_ARITH["color"], "color_fwd"), _lib.stream_ptr()),
                       "spf_color_forward")
        if acc is not None:      # the RBF-weighted mean meets up to four partial sums per entry: order-independent in this mode
            _fixed_flush(acc, agg3)
        ctx.arith = _ARITH["color"]
        if train:
            ctx.save_for_backward(wn, packed, *bufs)
            ctx.pl, ctx.NP, ctx.n_table = pl, NP, feat_col.shape[0]
            sinks = [_sink(t) for t in (feat_col, w0, b0, w2, b2, w4, b4)]
            ctx.sinks = sinks if (ctx.static and all(s_ is not None for s_ in sinks)) else None
        return agg3

    @staticmethod
    def backward(ctx, g_agg3):
        wn, packed, act0, act1, act2, masks = ctx.saved_tensors
        pl = ctx.pl
        dev = g_agg3.device
        rows = act1.shape[0]
        G1, G2, G3 = (torch.empty((rows, 256), dtype=torch.float32, device=dev) for _ in range(3))
        sk = ctx.sinks
        if sk is not None:                       # accumulate straight into the .grad buffers (set_grad_sinks)
            g_feat, g_b0, g_b2, g_b4 = sk[0], sk[2], sk[4], sk[6]
        else:
            g_bias = torch.zeros((3, 256), dtype=torch.float32, device=dev)
            g_b0, g_b2, g_b4 = g_bias[0], g_bias[1], g_bias[2]
            g_feat = torch.zeros((ctx.n_table, 64), dtype=torch.float32, device=dev)
        g_agg3 = g_agg3.contiguous()
        acc = _fixed_acc(g_feat, "color_latents") if (_SCATTER["mode"] == "fixed" and ctx.arith == 0) else None
        if _SCATTER["mode"] == "fixed" and ctx.arith != 0:
            raise RuntimeError("scatter mode 'fixed' needs the default ('split') colour kernels")
        with torch.cuda.device(dev), _prof.span("color_bwd", pairs=pl.n_pairs):
            _lib.check(_lib.lib().spf_color_backward(_lib.ptr(g_agg3), _lib.ptr(pl.nbr), _lib.ptr(wn), _lib.ptr(pl.point_slot), _lib.ptr(pl.pair_off),
                                                     _lib.ptr(pl.pair_point), _lib.ptr(pl.n_pairs), ctx.NP, pl.k, _lib.ptr(packed), _lib.ptr(masks),
                                                     _lib.ptr(G1), _lib.ptr(G2), _lib.ptr(G3), _lib.ptr(g_b0), _lib.ptr(g_b2), _lib.ptr(g_b4),
                                                     _lib.ptr(g_feat), _lib.ptr(acc), _arith_of(ctx.arith, "color_bwd"), _lib.stream_ptr()), "spf_color_backward")
        if acc is not None:
            _fixed_flush(acc, g_feat)
        if sk is not None:
            _bucket("color_latents")         # final: its all-reduce may overlap the weight-gradient GEMMs below
        # split-product kernels (the default) leave the bias gradients to the weight-gradient GEMM (column sums of G)
        kb = (lambda b: b) if ctx.arith == 0 else (lambda b: None)
        # the bf16-piece kernels write act1 / act2 as K-major 16-row blocks, G3 / G2 / G1 as K-major 64-row tiles (include/spurfies_hip.h: SPF_WGRAD_*)
        GT, AT, G64 = (WGRAD_G_TILES, WGRAD_A_TILES, WGRAD_G_TILES64) if ctx.arith == 0 else (0, 0, 0)
        if sk is not None:
            # layer 0's [256,104] product comes in the kernels' internal column order [latent 64 | encoding 39 | pad]: the reduce kernel
            # rotates it into the reference order [encoding 39 | latent 64] on the way (no permutation pass)
            if _ARITH["wgrad"] == 0 and ctx.arith == 0 and _TRUNK_BATCHED[0]:
                # the three GEMMs side by side in one launch + one reduce (round 4; they were three + three) — and with them the head stage's
                # three, deferred by RHead.backward (round 5)
                pend = [p for ps, _ in _PENDING_WGRAD for p in ps]
                if len(pend) + 3 > WGRAD_MAX_PROBLEMS:     # more than one head stage waiting (a model evaluated twice in one backward): own launch
                    flush_pending_wgrad()
                    pend = []
                _PENDING_WGRAD.clear()
                wgrad_batched([(G1, act0, sk[1], g_b0, 104, G64, 39, 103), (G2, act1, sk[3], g_b2, 256, G64 | AT, 0, 0),
                               (G3, act2, sk[5], g_b4, 256, G64 | AT, 0, 0)] + pend, pl.n_pairs)
                if pend:
                    _bucket("head")
            else:
                if _ARITH["wgrad"] == 0 and ctx.arith == 0:
                    wgrad(G1, act0, pl.n_pairs, out=sk[1], dbias=kb(g_b0), layout=G64, col_rot=39, col_mod=103)
                else:
                    sk[1].index_add_(1, _color_col_perm(dev), wgrad(G1, act0, pl.n_pairs, dbias=kb(g_b0), layout=G64)[:, :103])
                wgrad(G2, act1, pl.n_pairs, out=sk[3], dbias=kb(g_b2), layout=G64 | AT)
                wgrad(G3, act2, pl.n_pairs, out=sk[5], dbias=kb(g_b4), layout=G64 | AT)
            _bucket("color_weights")
            return (None,) * 14
        # exact-size (default) and worst-case (sync-free) buffers alike: the weight-gradient kernel reads the row count on the device
        if ctx.arith == 0 and _ARITH["wgrad"] == 0:
            dw0 = wgrad(G1, act0, pl.n_pairs, out=torch.zeros((256, 103), dtype=torch.float32, device=dev), dbias=kb(g_b0), layout=G64, col_rot=39, col_mod=103)
        else:
            dw0 = torch.empty((256, 103), dtype=torch.float32, device=dev)
            dw0[:, _color_col_perm(dev)] = wgrad(G1, act0, pl.n_pairs, dbias=kb(g_b0), layout=G64)[:, :103]   # [256,104] comes in the kernels' internal column order
        dw2, dw4 = wgrad(G2, act1, pl.n_pairs, dbias=kb(g_b2), layout=G64 | AT), wgrad(G3, act2, pl.n_pairs, dbias=kb(g_b4), layout=G64 | AT)
        grads = (g_feat, dw0, g_b0, dw2, g_b2, dw4, g_b4)
        return grads + (None,) * 7


# ---- per-ray compositing --------------------------------------------------------------------------
def filter_points(loc, slot_valid, cam_loc, ray_dirs):
    """pointneus_disent.py:207-239 on dense rows -> (z [R,SR], deltas [R,SR], x [R*SR,3]); no autograd (inputs are detached)."""
    R, SR = slot_valid.shape
    dev = loc.device
    z = torch.empty((R, SR), dtype=torch.float32, device=dev)
    deltas = torch.empty((R, SR), dtype=torch.float32, device=dev)
    x = torch.empty((R * SR, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_filter_points(_lib.ptr(loc.contiguous()), _lib.ptr(slot_valid), _lib.ptr(cam_loc.contiguous()),
                                                _lib.ptr(ray_dirs.contiguous()), R, SR, _lib.ptr(z), _lib.ptr(deltas), _lib.ptr(x),
                                                _lib.stream_ptr()), "spf_filter_points")
    return z, deltas, x


class Render(torch.autograd.Function):
    """(weights [R,SR], rgb [R,3], depth [R,1], dist_map [R], acc [R,1]) from sdf [R,SR], colors [R,SR,3] and the
    effective beta (a 0-dim tensor); differentiable w.r.t. all three."""

    @staticmethod
    def forward(ctx, sdf, colors, beta, slot_valid, z, deltas, beta_param=None, cam_loc=None, ray_dirs=None, local=None):
        """beta_param: the raw LaplaceDensity parameter (beta = |beta_param| + beta_min was formed from it, detached) when
        its gradient should be accumulated straight into its .grad buffer (set_grad_sinks).
        local (LocalTerms, optional): the feature-consistency term of this step (local_forward on the same sdf rows) — its gradient w.r.t.
        the SDF joins g_sdf inside the backward launch, scaled by the `lscale` the loss backward (FusedLoss, which runs first) leaves.
        cam_loc / ray_dirs [R,3] (optional): a sixth output pts_rendered = cam_loc + ray_dirs * dist (pointneus_disent.py:765-767; differentiable
        through dist) comes out of the same launch and its gradient goes back through the same backward launch."""
        R, SR = sdf.shape
        dev = sdf.device
        ctx.set_materialize_grads(False)        # backward handles None: no zero tensors (fill launches) for the outputs the loss ignores
        sdf_c, col_c = sdf.detach().contiguous(), colors.detach().contiguous()
        beta_c = beta.detach().reshape(1).contiguous()
        weights = torch.empty((R, SR), dtype=torch.float32, device=dev)
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
        dist = torch.empty((R,), dtype=torch.float32, device=dev)
        acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
        with_pts = cam_loc is not None
        pts = torch.empty((R, 3), dtype=torch.float32, device=dev) if with_pts else None
        dirs_c = ray_dirs.detach().contiguous() if with_pts else None
        loc_c = cam_loc.detach().contiguous() if with_pts else None
        with torch.cuda.device(dev), _prof.span("render_fwd", rays=R, slots=SR):
            _lib.check(_lib.lib().spf_render_forward(_lib.ptr(sdf_c), _lib.ptr(slot_valid), _lib.ptr(z), _lib.ptr(deltas), _lib.ptr(col_c),
                                                     _lib.ptr(beta_c), R, SR, _lib.ptr(weights), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(dist),
                                                     _lib.ptr(acc), _lib.ptr(loc_c), _lib.ptr(dirs_c), _lib.ptr(pts), None, None, None, 0.0, _lib.stream_ptr()),
                       "spf_render_forward")
        ctx.save_for_backward(sdf_c, col_c, beta_c, slot_valid, z, deltas, weights, dirs_c)
        ctx.with_pts = with_pts
        ctx.local = local
        ctx.beta_param = beta_param.detach() if beta_param is not None else None
        ctx.beta_sink = _sink(beta_param) if beta_param is not None else None
        if beta_param is not None and ctx.beta_sink is None:
            raise RuntimeError("Render: beta_param needs a gradient sink (ops.set_grad_sinks)")
        return (weights, rgb, depth, dist, acc, pts) if with_pts else (weights, rgb, depth, dist, acc)

    @staticmethod
    def backward(ctx, g_w, g_rgb, g_depth, g_dist, g_acc, g_pts=None):
        sdf, colors, beta, slot_valid, z, deltas, weights, dirs = ctx.saved_tensors
        R, SR = sdf.shape
        dev = sdf.device
        gw = None if g_w is None else g_w.contiguous()
        g_acc = None if g_acc is None else g_acc.reshape(R).contiguous()     # acc = sum_j w_j: added to every slot's gradient inside the kernel
        g_pts = None if (g_pts is None or not ctx.with_pts) else g_pts.contiguous()
        g_rgb = torch.zeros((R, 3), device=dev) if g_rgb is None else g_rgb.contiguous()
        g_depth = None if g_depth is None else g_depth.contiguous()
        g_dist = None if g_dist is None else g_dist.contiguous()
        g_sdf = torch.empty((R, SR), dtype=torch.float32, device=dev)
        g_col = torch.empty((R, SR, 3), dtype=torch.float32, device=dev)
        sink = ctx.beta_sink
        g_beta = sink.reshape(1) if sink is not None else torch.zeros((1,), dtype=torch.float32, device=dev)
        acc_b = _fixed_acc(g_beta, "beta") if _SCATTER["mode"] == "fixed" else None        # one term per ray: order-independent in this mode
        lt = ctx.local
        if lt is not None and not lt.scaled:
            raise RuntimeError("Render(local=...): the loss backward (FusedLoss with the same LocalTerms) has not written lscale")
        with torch.cuda.device(dev), _prof.span("render_bwd", rays=R, slots=SR):
            _lib.check(_lib.lib().spf_render_backward(_lib.ptr(sdf), _lib.ptr(slot_valid), _lib.ptr(z), _lib.ptr(deltas), _lib.ptr(colors),
                                                      _lib.ptr(beta), _lib.ptr(weights), _lib.ptr(gw), _lib.ptr(g_rgb), _lib.ptr(g_depth),
                                                      _lib.ptr(g_dist), R, SR, _lib.ptr(g_sdf), _lib.ptr(g_col), _lib.ptr(g_beta),
                                                      _lib.ptr(ctx.beta_param), _lib.ptr(g_acc), _lib.ptr(g_pts), _lib.ptr(dirs if g_pts is not None else None),
                                                      _lib.ptr(acc_b), _lib.ptr(None if lt is None else lt.lfirst), _lib.ptr(None if lt is None else lt.lcoef),
                                                      _lib.ptr(None if lt is None else lt.lscale), _lib.stream_ptr()), "spf_render_backward")
        if acc_b is not None:
            _fixed_flush(acc_b, g_beta)
        return g_sdf, g_col, (None if sink is not None else g_beta.reshape(())), None, None, None, None, None, None, None


def render_eval(sdf, colors, beta, slot_valid, z, deltas, grad, ray_valid, depth_fill=1.0):
    """The evaluation render's composites in one launch (no autograd): (weights [R,SR], rgb [R,3], depth_values [R,1] — `depth_fill` on rays
    without a valid slot —, normal_map [R,3] = sum_j w_j grad_j / |grad_j|): pointneus_disent.py:714-723, 782-815, 822-826, 886-888."""
    R, SR = sdf.shape
    dev = sdf.device
    weights = torch.empty((R, SR), dtype=torch.float32, device=dev)
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
    normal = torch.empty((R, 3), dtype=torch.float32, device=dev)
    scratch = torch.empty((2, R), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _prof.span("render_fwd", rays=R, slots=SR):
        _lib.check(_lib.lib().spf_render_forward(_lib.ptr(sdf.detach().contiguous()), _lib.ptr(slot_valid), _lib.ptr(z), _lib.ptr(deltas),
                                                 _lib.ptr(colors.detach().contiguous()), _lib.ptr(beta.detach().reshape(1).contiguous()), R, SR, _lib.ptr(weights),
                                                 _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(scratch[0]), _lib.ptr(scratch[1]), None, None, None,
                                                 _lib.ptr(grad.detach().contiguous()), _lib.ptr(normal), _lib.ptr(ray_valid), float(depth_fill), _lib.stream_ptr()),
                   "spf_render_forward")
    return weights, rgb, depth, normal


class RenderW(torch.autograd.Function):
    """The part of `Render` that depends on the SDF alone: (weights [R,SR], depth [R,1], dist_map [R], acc [R,1], pts_rendered [R,3]) —
    spf_render_forward's weights-only form.  With `RenderRGB` behind the colour MLPs it replaces `Render` in a forked optimisation step:
    the pseudo-point pass (which needs pts_rendered) then runs beside the colour stage, and in the backward the colour branch does not wait
    for the pseudo-point pass's gradient.  Same sums as the fused form (tests/test_gpu_render.py)."""

    @staticmethod
    def forward(ctx, sdf, beta, slot_valid, z, deltas, beta_param, cam_loc, ray_dirs):
        R, SR = sdf.shape
        dev = sdf.device
        ctx.set_materialize_grads(False)
        sdf_c, beta_c = sdf.detach().contiguous(), beta.detach().reshape(1).contiguous()
        weights = torch.empty((R, SR), dtype=torch.float32, device=dev)
        depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
        dist = torch.empty((R,), dtype=torch.float32, device=dev)
        acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
        pts = torch.empty((R, 3), dtype=torch.float32, device=dev)
        dirs_c, loc_c = ray_dirs.detach().contiguous(), cam_loc.detach().contiguous()
        with torch.cuda.device(dev), _prof.span("render_fwd", rays=R, slots=SR):
            _lib.check(_lib.lib().spf_render_forward(_lib.ptr(sdf_c), _lib.ptr(slot_valid), _lib.ptr(z), _lib.ptr(deltas), None, _lib.ptr(beta_c), R, SR,
                                                     _lib.ptr(weights), None, _lib.ptr(depth), _lib.ptr(dist), _lib.ptr(acc), _lib.ptr(loc_c),
                                                     _lib.ptr(dirs_c), _lib.ptr(pts), None, None, None, 0.0, _lib.stream_ptr()), "spf_render_forward")
        ctx.save_for_backward(sdf_c, beta_c, slot_valid, z, deltas, weights, dirs_c)
        ctx.beta_param = beta_param.detach()
        ctx.beta_sink = _sink(beta_param)
        if ctx.beta_sink is None:
            raise RuntimeError("RenderW: beta_param needs a gradient sink (ops.set_grad_sinks)")
        return weights, depth, dist, acc, pts

    @staticmethod
    def backward(ctx, g_w, g_depth, g_dist, g_acc, g_pts):
        sdf, beta, slot_valid, z, deltas, weights, dirs = ctx.saved_tensors
        R, SR = sdf.shape
        dev = sdf.device
        c = lambda t: None if t is None else t.contiguous()
        g_w, g_depth, g_dist, g_pts = c(g_w), c(g_depth), c(g_dist), c(g_pts)
        g_acc = None if g_acc is None else g_acc.reshape(R).contiguous()
        g_sdf = torch.empty((R, SR), dtype=torch.float32, device=dev)
        g_beta = ctx.beta_sink.reshape(1)
        acc_b = _fixed_acc(g_beta, "beta") if _SCATTER["mode"] == "fixed" else None
        with torch.cuda.device(dev), _prof.span("render_bwd", rays=R, slots=SR):
            _lib.check(_lib.lib().spf_render_backward(_lib.ptr(sdf), _lib.ptr(slot_valid), _lib.ptr(z), _lib.ptr(deltas), None, _lib.ptr(beta),
                                                      _lib.ptr(weights), _lib.ptr(g_w), None, _lib.ptr(g_depth), _lib.ptr(g_dist), R, SR, _lib.ptr(g_sdf),
                                                      None, _lib.ptr(g_beta), _lib.ptr(ctx.beta_param), _lib.ptr(g_acc), _lib.ptr(g_pts),
                                                      _lib.ptr(dirs if g_pts is not None else None), _lib.ptr(acc_b), None, None, None, _lib.stream_ptr()),
                       "spf_render_backward")
        if acc_b is not None:
            _fixed_flush(acc_b, g_beta)
        return g_sdf, None, None, None, None, None, None, None


class RenderRGB(torch.autograd.Function):
    """rgb [R,3] = sum_j weights[r,j] colors[r,j,:] (spf_render_rgb; backward: g_colors = weights g_rgb, g_weights = colors . g_rgb)."""

    @staticmethod
    def forward(ctx, weights, colors):
        R, SR = weights.shape
        dev = weights.device
        w_c, col_c = weights.detach().contiguous(), colors.detach().contiguous()
        rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_render_rgb(_lib.ptr(w_c), _lib.ptr(col_c), R, SR, _lib.ptr(rgb), _lib.stream_ptr()), "spf_render_rgb")
        ctx.save_for_backward(w_c, col_c)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        w, col = ctx.saved_tensors
        R, SR = w.shape
        dev = w.device
        g_rgb = g_rgb.contiguous()
        g_col = torch.empty((R, SR, 3), dtype=torch.float32, device=dev)
        g_w = torch.empty((R, SR), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_render_rgb_backward(_lib.ptr(w), _lib.ptr(col), _lib.ptr(g_rgb), R, SR, _lib.ptr(g_col), _lib.ptr(g_w),
                                                          _lib.stream_ptr()), "spf_render_rgb_backward")
        return g_w, g_col


# ---- sampler stages ---------------------------------------------------------------------------------
def sampler_uniform(tlin, t_rand, cam_loc, ray_dirs, near, far):
    R, n = cam_loc.shape[0], tlin.shape[0]
    dev = cam_loc.device
    z = torch.empty((R, n), dtype=torch.float32, device=dev)
    pts = torch.empty((R, n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_sampler_uniform(_lib.ptr(tlin), _lib.ptr(t_rand), _lib.ptr(cam_loc), _lib.ptr(ray_dirs), R, n, float(near),
                                                  float(far), _lib.ptr(z), _lib.ptr(pts), _lib.stream_ptr()), "spf_sampler_uniform")
    return z, pts


def sampler_iter(z, sdf, beta_in, beta0, eps, bound_coef, beta_iters, more, add_tiny, u, N, flags=None, it=0, beta_out=None):
    """-> (samples [R,N] | None, beta [R], z_merged [R,n+N] | None, merged_idx [R,n+N] | None).
    flags / it: device-side loop control (include/spurfies_hip.h: spf_sampler_iter) — a call whose turn it is not leaves its outputs
    untouched, so with flags they are ZERO-initialised (a later gather must find valid indices) and beta_out may be passed in."""
    R, n = z.shape
    dev = z.device
    new = torch.empty if flags is None else torch.zeros
    samples = new((R, N), dtype=torch.float32, device=dev) if N > 0 else None
    beta = beta_out if beta_out is not None else new((R,), dtype=torch.float32, device=dev)
    zm = new((R, n + N), dtype=torch.float32, device=dev) if more else None
    mi = new((R, n + N), dtype=torch.int32, device=dev) if more else None
    per_ray = 1 if (u is not None and u.dim() == 2) else 0
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_sampler_iter(_lib.ptr(z), _lib.ptr(sdf), _lib.ptr(beta_in), _lib.ptr(beta0), R, n, float(eps), float(bound_coef),
                                               int(beta_iters), int(bool(more)), float(add_tiny), _lib.ptr(u), per_ray, N, _lib.ptr(samples),
                                               _lib.ptr(beta), _lib.ptr(zm), _lib.ptr(mi), _lib.ptr(flags), int(it), _lib.stream_ptr()), "spf_sampler_iter")
    return samples, beta, zm, mi


def sampler_train(z, pair_tmp, pl, beta0, eps, bound_coef, beta_iters, u, sel, near, far, cam_loc, ray_dirs, grid_handle, SR):
    """The optimisation step's sampler pass behind the SDF kernel as ONE launch (include/spurfies_hip.h: spf_sampler_train): per-sample SDF from
    the geometry kernel's per-pair scratch, beta + inverse-CDF samples, the sorted z / main-pass points and the main pass's slot assignment.
    -> (beta [R], z_out [R,M], points [R,M,3], slot_sample int32 [R,SR], ray_valid uint8 [R] (cleared))."""
    R, n = z.shape
    N = u.shape[1]
    Ne = 0 if sel is None else sel.shape[0]
    M = N + 2 + Ne
    dev = z.device
    beta = torch.empty((R,), dtype=torch.float32, device=dev)
    z_out = torch.empty((R, M), dtype=torch.float32, device=dev)
    pts = torch.empty((R, M, 3), dtype=torch.float32, device=dev)
    slot_sample = torch.empty((R, SR), dtype=torch.int32, device=dev)
    ray_valid = torch.empty((R,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_sampler_train(_lib.ptr(z), _lib.ptr(pair_tmp), _lib.ptr(pl.pair_off), _lib.ptr(pl.slot_point), _lib.ptr(beta0), R, n, float(eps),
                                                float(bound_coef), int(beta_iters), _lib.ptr(u), N, _lib.ptr(sel), Ne, float(near), float(far), _lib.ptr(cam_loc),
                                                _lib.ptr(ray_dirs), grid_handle, int(SR), _lib.ptr(beta), _lib.ptr(z_out), _lib.ptr(pts), _lib.ptr(slot_sample),
                                                _lib.ptr(ray_valid), _lib.stream_ptr()), "spf_sampler_train")
    return beta, z_out, pts, slot_sample, ray_valid


def sampler_eval(phase, grid_handle, R, **kw):
    """One launch of the evaluation sampler loop's iteration (include/spurfies_hip.h: spf_sampler_eval; phase 0 = test, 1 = step, 2 = last).
    kw: the fields of spf_sampler_eval_args — tensors (or None) for the pointers, numbers for the rest."""
    a = _lib.SamplerEvalArgs()
    dev = kw["z"].device
    for name, ctype in _lib.SamplerEvalArgs._fields_:
        v = kw.get(name)
        if ctype is ctypes.c_void_p:
            setattr(a, name, None if v is None else v.data_ptr())
        elif v is not None:
            setattr(a, name, v)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_sampler_eval(ctypes.byref(a), grid_handle, int(R), int(phase), _lib.stream_ptr()), "spf_sampler_eval")


def sampler_finish(z_samples, z_vals, sel, near, far, cam_loc, ray_dirs, flags=None, it=0, out=None):
    """flags / it: runs only behind the final sampling pass of iteration `it` (spf_sampler_iter); out = (z_out, points) to write into."""
    R, Ns = z_samples.shape
    n, Ne = z_vals.shape[1], (0 if sel is None else sel.shape[0])
    dev = z_samples.device
    M = Ns + 2 + Ne
    z_out, pts = out if out is not None else (torch.empty((R, M), dtype=torch.float32, device=dev), torch.empty((R, M, 3), dtype=torch.float32, device=dev))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_sampler_finish(_lib.ptr(z_samples), Ns, _lib.ptr(z_vals), n, _lib.ptr(sel), Ne, float(near), float(far),
                                                 _lib.ptr(cam_loc), _lib.ptr(ray_dirs), R, _lib.ptr(z_out), _lib.ptr(pts), _lib.ptr(flags), int(it),
                                                 _lib.stream_ptr()), "spf_sampler_finish")
    return z_out, pts


# ---- radiance head R -------------------------------------------------------------------------------
def pack_rhead_weights(ws, zero=None):
    """ws: [F_color.6 w, b, R.0 w, b, R.2 w, b, R.4 w, b] -> packed image of the head stage.  zero: a float32 buffer cleared by the same launch."""
    args = [t.detach().contiguous().float() for t in ws]
    dev = args[0].device
    packed = torch.empty((int(_lib.lib().spf_rhead_packed_floats()),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_rhead_pack(*[_lib.ptr(a) for a in args], _lib.ptr(packed), _lib.ptr(zero), 0 if zero is None else zero.numel(),
                                             _lib.stream_ptr()), "spf_rhead_pack")
    return packed


class RHead(_GradModeFunction):
    """colors [rows,3] = sigmoid(R([direnc3(ray dir) | F_color.6(agg3)])) on the P valid points (pointneus_disent.py:333-346),
    written at the points' slot rows (0 elsewhere).  F_color's linear last layer is applied here, per point, to the
    RBF-weighted mean `agg3` that `ColorAgg` produced.  Forward and the data-gradient chain are HIP kernels; the wide
    layers' weight gradients are spf_wgrad launches over [P,256] buffers."""

    @staticmethod
    def forward(ctx, agg3, w6, b6, w0, b0, w2, b2, w4, b4, ray_dirs, point_slot, n_points, SR, n_rows, static=False, pre=None):
        """agg3 has exactly P rows (host-known), or — static=True — worst-case rows with the count read on the device.
        pre: (packed image, zeroed colors [n_rows,3]) from prepack_rhead."""
        dev = agg3.device
        ctx.static = static
        P = agg3.shape[0]
        tiles = (P + 63) // 64
        T = 64 * tiles
        if pre is not None and pre[1].shape[0] == n_rows:
            packed, colors = pre
        else:
            colors = torch.empty((n_rows, 3), dtype=torch.float32, device=dev)      # cleared by the packing launch
            packed = pack_rhead_weights([w6, b6, w0, b0, w2, b2, w4, b4], zero=colors)
        train = _GradModeFunction._outer_grad_mode and any(ctx.needs_input_grad[:9])
        if train:
            bufs = [torch.empty((T, 256), dtype=torch.float32, device=dev), torch.empty((T, 24), dtype=torch.float32, device=dev),
                    torch.empty((T, 256), dtype=torch.float32, device=dev), torch.empty((T, 256), dtype=torch.float32, device=dev),
                    torch.empty((tiles, 2, 512), dtype=torch.int32, device=dev)]
        else:
            bufs = [None] * 5
        agg3_c = agg3.detach().contiguous()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_rhead_forward(_lib.ptr(agg3_c), _lib.ptr(ray_dirs), _lib.ptr(point_slot), _lib.ptr(n_points), P, int(SR),
                                                    _lib.ptr(packed), _lib.ptr(colors), *[_lib.ptr(b) for b in bufs], _arith_of(_ARITH["rhead"], "rhead_fwd"),
                                                    _lib.stream_ptr()), "spf_rhead_forward")
        ctx.arith = _ARITH["rhead"]
        if train:
            ctx.save_for_backward(agg3_c, colors, point_slot, n_points, packed, *bufs)
            sinks = [_sink(t) for t in (w6, b6, w0, b0, w2, b2, w4, b4)]
            ctx.sinks = sinks if (static and all(s_ is not None for s_ in sinks)) else None
        return colors

    @staticmethod
    def backward(ctx, g_colors):
        agg3, colors, point_slot, n_points, packed, agg, direnc, act1, act2, masks = ctx.saved_tensors
        dev = g_colors.device
        P, T = agg3.shape[0], act1.shape[0]
        G1, G2, g_agg, g_agg3 = (torch.empty((T, 256), dtype=torch.float32, device=dev) for _ in range(4))
        sk = ctx.sinks
        if sk is not None:
            g_b6, g_b0, g_b2, g_w4, g_b4 = sk[1], sk[3], sk[5], sk[6], sk[7]
        else:
            g_small = torch.zeros((1539,), dtype=torch.float32, device=dev)
            g_b6, g_b0, g_b2, g_w4, g_b4 = (g_small[:256], g_small[256:512], g_small[512:768], g_small[768:1536].view(3, 256),
                                            g_small[1536:1539])
        g_colors = g_colors.contiguous()
        fixed = _SCATTER["mode"] == "fixed" and ctx.arith == 0
        acc_w, acc_b = (_fixed_acc(g_w4, "r4_weight"), _fixed_acc(g_b4, "r4_bias")) if fixed else (None, None)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_rhead_backward(_lib.ptr(g_colors), _lib.ptr(colors), _lib.ptr(point_slot), _lib.ptr(n_points), P, _lib.ptr(packed),
                                                     _lib.ptr(act2), _lib.ptr(masks), _lib.ptr(G1), _lib.ptr(G2), _lib.ptr(g_agg), _lib.ptr(g_agg3),
                                                     _lib.ptr(g_b6), _lib.ptr(g_b0), _lib.ptr(g_b2), _lib.ptr(g_w4), _lib.ptr(g_b4), _lib.ptr(acc_w),
                                                     _lib.ptr(acc_b), _arith_of(ctx.arith, "rhead_bwd"), _lib.stream_ptr()), "spf_rhead_backward")
        if fixed:
            _fixed_flush(acc_w, g_w4)
            _fixed_flush(acc_b, g_b4)
        # split-product kernels (the default) leave the 256-wide layers' bias gradients to the weight-gradient GEMMs (column sums)
        split = ctx.arith == 0
        kb = (lambda b: b) if split else (lambda b: None)
        if sk is not None:
            # F_color.6 (K = points, not pairs), R.0's agg block (reference column order [dir-enc | agg]) and R.2: side by side — and, in a
            # forked step, on a branch of their own: nothing behind them on this stream (the colour backward) reads what they write
            probs = [(g_agg, agg3, sk[0], kb(g_b6)), (G1, agg, sk[2][:, 21:], kb(g_b0)), (G2, act1, sk[4], kb(g_b2))]
            if _MERGE_HEAD[0] and split and _ARITH["wgrad"] == 0 and _ARITH["color"] == 0 and _TRUNK_BATCHED[0] and _SCATTER["mode"] != "fixed":
                # round 5: these three GEMMs (K = valid points: a dozen workgroups' worth of rows at 128 rays) ride in the colour trunk's batched
                # launch, which autograd issues next (ColorAgg.backward) — one pipeline ramp / tail / slab reduce for all six
                rows = min(g_agg.shape[0], agg3.shape[0])
                # ... and with them R.0's 21 view-encoding columns (direnc [rows,24], its three padding columns are zero): a fourth, narrow problem
                _PENDING_WGRAD.append(([pr + (256, 0, 0, 0, n_points, rows) for pr in probs] + [(G1, direnc, sk[2], None, 24, 0, 0, 21, n_points, rows)],
                                       (g_agg, agg3, G1, agg, G2, act1, direnc, n_points)))
            else:
                with branch("wgrad_head", dev):
                    wgrad_batched(probs, n_points)
                    wgrad(G1, direnc, n_points, C=21, out=sk[2])
                    keep(g_agg, agg3, G1, agg, G2, act1, direnc, n_points)
                    _bucket("head")              # F_color.6, R.*, and density.beta (written by the compositing backward before this node ran)
            return (g_agg3[:P],) + (None,) * 15
        dw6 = wgrad(g_agg, agg3, n_points, dbias=kb(g_b6))
        dw0 = torch.zeros((256, 277), dtype=torch.float32, device=dev)        # reference column order [dir-enc | agg]
        wgrad(G1, direnc, n_points, C=21, out=dw0)
        wgrad(G1, agg, n_points, out=dw0[:, 21:], dbias=kb(g_b0))
        dw2 = wgrad(G2, act1, n_points, dbias=kb(g_b2))
        return (g_agg3[:P], dw6, g_b6, dw0, g_b0, dw2, g_b2, g_w4, g_b4, None, None, None, None, None, None, None)


_wgrad_ws = {}
_TRUNK_BATCHED = [True]


def set_trunk_wgrad_batched(on=True):
    """The colour trunk's three weight-gradient GEMMs as one side-by-side launch (default) or one after the other (tests / A-B runs)."""
    _TRUNK_BATCHED[0] = bool(on)


def set_color_mode(mode: str):
    """'split' (default) or 'f32' for the colour trunk kernels; a backward runs in the arithmetic its forward recorded."""
    _ARITH["color"] = _ARITH_NAMES[mode]


def color_mode() -> str:
    return "split" if _ARITH["color"] == 0 else "f32"


def set_rhead_mode(mode: str):
    """'split' (default) or 'f32' for the per-point head kernels."""
    _ARITH["rhead"] = _ARITH_NAMES[mode]


def rhead_mode() -> str:
    return "split" if _ARITH["rhead"] == 0 else "f32"


def set_wgrad_mode(mode: str):
    """'split' (default) or 'f32' for the weight-gradient GEMMs."""
    _ARITH["wgrad"] = _ARITH_NAMES[mode]



WGRAD_G_TILES, WGRAD_A_TILES, WGRAD_G_TILES64 = 1, 2, 4      # spf_wgrad layout bits: operand stored as K-major blocks [block][256 features][16 | 64 rows]
WGRAD_DETERMINISTIC = 8                                       # fixed-order slab / column-sum reduce (set_scatter_mode("fixed"))


def _wgrad_det(C):
    """The deterministic reduce is part of scatter mode 'fixed' (bit-reproducible steps); wide operands need the default arithmetic for it."""
    return WGRAD_DETERMINISTIC if (_SCATTER["mode"] == "fixed" and (_ARITH["wgrad"] == 0 or C <= 32)) else 0


def wgrad(G, A, n_rows, C=None, out=None, ldw=None, dbias=None, layout=0, col_rot=0, col_mod=0):
    """out[256, :C] += G[:rows]^T A[:rows, :C] with the row count `n_rows` (int32 device tensor or None) read on the device;
    dbias [256] (optional) += column sums of G[:rows] (the same layer's bias gradient).  col_rot / col_mod: product column i < col_mod is
    accumulated into out column (i + col_rot) % col_mod (include/spurfies_hip.h: spf_wgrad)."""
    dev = G.device
    C = A.shape[1] if C is None else C
    if out is None:
        out = torch.zeros((256, C), dtype=torch.float32, device=dev)
    ldw = out.stride(0) if ldw is None else ldw
    nws = int(_lib.lib().spf_wgrad_workspace_floats(C))
    key = _scratch_key(dev) + (nws,)      # scenes stepped at the same time must not share scratch
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.empty((nws,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_wgrad(_lib.ptr(G), _lib.ptr(A), A.stride(0), C, _lib.ptr(n_rows), min(G.shape[0], A.shape[0]), _lib.ptr(out), ldw, _lib.ptr(dbias),
                                        _lib.ptr(_wgrad_ws[key]), int(layout) | _wgrad_det(C), _arith_of(_ARITH["wgrad"], "wgrad") if C > 32 else _ARITH["wgrad"],
                                        int(col_rot), int(col_mod), _lib.stream_ptr()),
                   "spf_wgrad")
    return out


_PENDING_WGRAD = []          # (problems with their own row counts, tensors to keep alive): the head stage's GEMMs, waiting for the trunk's launch
_MERGE_HEAD = [True]


def set_head_wgrad_merged(on=True):
    """The head stage's three 256-wide weight-gradient GEMMs ride in the colour trunk's batched launch (default) or run as a launch pair of
    their own right behind the head's backward kernel (tests / A-B runs)."""
    _MERGE_HEAD[0] = bool(on)


WGRAD_MAX_PROBLEMS = 7       # spf_wgrad_batched (include/spurfies_hip.h)


def flush_pending_wgrad():
    """Launch weight-gradient problems that were deferred to a later batched launch which never came (a backward without the colour trunk)."""
    while _PENDING_WGRAD:
        probs, _ = _PENDING_WGRAD.pop(0)
        wgrad_batched(probs, None)
        _bucket("head")


def drop_pending_wgrad():
    """Forget deferred weight-gradient problems WITHOUT launching them: called where a new forward starts — anything still waiting belongs to a
    backward that was abandoned half way (an exception between the head's and the trunk's backward) and must not ride in the next step's launch."""
    _PENDING_WGRAD.clear()


def wgrad_batched(problems, n_rows):
    """problems: up to six (G, A, out, dbias | None[, C, layout, col_rot, col_mod[, own_n_rows, own_max_rows]]): out[256, :C] += G[:rows]^T A[:rows, :C],
    dbias += column sums of G[:rows], all in ONE pair of launches, side by side on the chip (spf_wgrad_batched).  Short tuples are row-major
    [rows,256] operands; the long form carries spf_wgrad's layout / column rotation per problem (the colour trunk's tiled operands) and,
    optionally, the problem's OWN row count (device int32 tensor or None, and the buffer's row capacity) instead of the call's `n_rows`."""
    dev = problems[0][0].device
    arr = (_lib.WgradProblem * len(problems))()
    max_rows = None
    work = []                # (row count: device tensor | None = the call's, capacity, C) per problem — read by the profiler span only
    for q, prob in enumerate(problems):
        G, A, out, dbias = prob[:4]
        rest = tuple(prob[4:])
        C, layout, col_rot, col_mod = (rest[:4] + (256, 0, 0, 0)[len(rest[:4]):]) if rest else (256, 0, 0, 0)
        own = rest[4:6] if len(rest) >= 6 else None
        if layout == 0 and C == 256 and (A.shape[1] != 256 or G.shape[1] != 256 or out.shape != (256, 256)):
            raise ValueError("wgrad_batched: a row-major problem is [rows,256]^T x [rows,256] -> [256,256] (or C <= 128 columns of a narrower A)")
        arr[q].G, arr[q].A, arr[q].lda = _lib.ptr(G), _lib.ptr(A), A.stride(0)
        arr[q].dW, arr[q].ldw, arr[q].dbias = _lib.ptr(out), out.stride(0), _lib.ptr(dbias)
        arr[q].C, arr[q].layout, arr[q].col_rot, arr[q].col_mod = int(C), int(layout), int(col_rot), int(col_mod)
        rows = min(G.shape[0], A.shape[0])
        work.append((own[0] if own is not None else None, int(min(rows, own[1])) if own is not None else rows, int(C)))
        if own is not None:
            arr[q].n_rows, arr[q].max_rows = _lib.ptr(own[0]), int(min(rows, own[1]))
        else:
            arr[q].n_rows, arr[q].max_rows = None, 0
            max_rows = rows if max_rows is None else min(max_rows, rows)
    nws = int(_lib.lib().spf_wgrad_workspace_floats(256)) * len(problems)
    key = _scratch_key(dev) + (nws,)
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.empty((nws,), dtype=torch.float32, device=dev)
    meta = {}
    if _prof.active():       # sum over the problems of rows x (256 + C) (operand floats read) and rows x C (x 512 = FLOP)
        cnt = [(torch.clamp((n if n is not None else n_rows).reshape(-1)[:1].double(), max=cap) if (n is not None or n_rows is not None) else
                torch.tensor([float(cap)], dtype=torch.float64, device=dev), C) for n, cap, C in work]
        meta = {"rows_x_c": sum(r * C for r, C in cnt), "operand_floats": sum(r * (256 + C) for r, C in cnt), "problems": len(problems)}
    with torch.cuda.device(dev), _prof.span("wgrad", **meta):
        _lib.check(_lib.lib().spf_wgrad_batched(arr, len(problems), _lib.ptr(n_rows), 0 if max_rows is None else max_rows, _lib.ptr(_wgrad_ws[key]), _arith_of(_ARITH["wgrad"], "wgrad"),
                                                    _wgrad_det(256), _lib.stream_ptr()),
                   "spf_wgrad_batched")


# ---- ray set-up and loss terms (one launch each instead of dozens of elementwise PyTorch kernels) ----------------
def camera_rays(uv, pose, intrinsics, beta_param=None, beta_min=0.0, beta_out=None):
    """uv [1,R,2], pose [1,4,4], intrinsics [1,3|4,3|4] -> (ray_dirs [R,3], cam_loc [R,3], depth_scale [R,1]) — the two
    rend_util.get_camera_params calls of pointneus_disent.py:640-650.  None when the batch holds several views or
    quaternion poses (the caller then keeps the PyTorch formulation).  beta_param / beta_min / beta_out (device scalars): the same launch
    writes the forward's effective Laplace scale |beta_param| + beta_min into beta_out (density.py:28-30)."""
    if uv.dim() != 3 or uv.shape[0] != 1 or pose.shape[-2:] != (4, 4) or not uv.is_cuda:
        return None
    R, dev = uv.shape[1], uv.device
    uv_c = uv.detach().reshape(R, 2).float().contiguous()
    pose_c = pose.detach().reshape(4, 4).float().contiguous()
    K = intrinsics.detach().float()
    ks = K.shape[-1]
    K = K.reshape(ks, ks).contiguous()
    dirs = torch.empty((R, 3), dtype=torch.float32, device=dev)
    loc = torch.empty((R, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((R, 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_camera_rays(_lib.ptr(uv_c), _lib.ptr(pose_c), _lib.ptr(K), ks, R, _lib.ptr(dirs), _lib.ptr(loc),
                                              _lib.ptr(scale), _lib.ptr(None if beta_out is None else beta_param.detach()), float(beta_min),
                                              _lib.ptr(beta_out), _lib.stream_ptr()), "spf_camera_rays")
    return dirs, loc, scale


def camera_uniform(uv, pose, intrinsics, beta_param, beta_min, beta_out, tlin, t_rand, near, far, tv_graph=None, tv_feat=None, pack=None):
    """camera_rays + sampler_uniform in one launch (spf_camera_uniform) -> (ray_dirs, cam_loc, depth_scale, z [R,n], points [R,n,3], tv | None);
    None for multi-view batches / quaternion poses (as camera_rays).  tv_graph (model.utils.TVGraph) + tv_feat [n,32]: the per-point TV terms
    ride in the same launch (values only: their backward rides in the loss backward launch, FusedLoss(tv_ctx=...)).
    pack = (F_color, R, n_rows): both weight-packing jobs ride along too; a seventh result `pre` = ((colour image, zeroed agg3 [n_rows,256]),
    (head image, zeroed colors [n_rows,3])) for ColorAgg / RHead (the step prologue: everything that reads only parameters and the batch)."""
    if uv.dim() != 3 or uv.shape[0] != 1 or pose.shape[-2:] != (4, 4) or not uv.is_cuda:
        return None
    R, dev, n = uv.shape[1], uv.device, tlin.shape[0]
    uv_c = uv.detach().reshape(R, 2).float().contiguous()
    pose_c = pose.detach().reshape(4, 4).float().contiguous()
    K = intrinsics.detach().float()
    ks = K.shape[-1]
    K = K.reshape(ks, ks).contiguous()
    dirs = torch.empty((R, 3), dtype=torch.float32, device=dev)
    loc = torch.empty((R, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((R, 1), dtype=torch.float32, device=dev)
    z = torch.empty((R, n), dtype=torch.float32, device=dev)
    pts = torch.empty((R, n, 3), dtype=torch.float32, device=dev)
    tv, g = None, tv_graph
    feat_c = None
    if g is not None:
        feat_c = tv_feat.detach().contiguous()
        tv = torch.empty((feat_c.shape[0],), dtype=torch.float32, device=dev)
    pk, pre, hold = None, None, None
    if pack is not None:
        import ctypes

        fc, rh, n_rows = pack
        ws = [t.detach().contiguous().float() for t in (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias,
                                                         fc[6].weight, fc[6].bias, rh[0].weight, rh[0].bias, rh[2].weight, rh[2].bias, rh[4].weight, rh[4].bias)]
        c_packed = torch.empty((int(_lib.lib().spf_color_packed_floats()),), dtype=torch.float32, device=dev)
        r_packed = torch.empty((int(_lib.lib().spf_rhead_packed_floats()),), dtype=torch.float32, device=dev)
        agg3 = torch.empty((n_rows, 256), dtype=torch.float32, device=dev)
        colors = torch.empty((n_rows, 3), dtype=torch.float32, device=dev)
        pk = _lib.ProloguePacks()
        for name, t in zip(("cw0", "cb0", "cw2", "cb2", "cw4", "cb4", "rw6", "rb6", "rw0", "rb0", "rw2", "rb2", "rw4", "rb4"), ws):
            setattr(pk, name, t.data_ptr())
        pk.c_packed, pk.c_zero, pk.c_zero_floats = c_packed.data_ptr(), agg3.data_ptr(), agg3.numel()
        pk.r_packed, pk.r_zero, pk.r_zero_floats = r_packed.data_ptr(), colors.data_ptr(), colors.numel()
        pre, hold = ((c_packed, agg3), (r_packed, colors)), ws
        pk = ctypes.byref(pk)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().spf_camera_uniform(_lib.ptr(uv_c), _lib.ptr(pose_c), _lib.ptr(K), ks, R, _lib.ptr(dirs), _lib.ptr(loc), _lib.ptr(scale),
                                                 _lib.ptr(None if beta_out is None else beta_param.detach()), float(beta_min), _lib.ptr(beta_out), _lib.ptr(tlin),
                                                 _lib.ptr(t_rand), n, float(near), float(far), _lib.ptr(z), _lib.ptr(pts), _lib.ptr(feat_c),
                                                 _lib.ptr(None if g is None else g.nbr), _lib.ptr(None if g is None else g.w), _lib.ptr(None if g is None else g.norm),
                                                 0 if g is None else feat_c.shape[0], 1 if g is None else g.nbr.shape[1], _lib.ptr(tv), pk, _lib.stream_ptr()),
                   "spf_camera_uniform")
    del hold
    return dirs, loc, scale, z, pts, tv, pre


_FUSED_SAMPLER = [True]


def set_fused_sampler(on=True):
    """The optimisation step's sampler chain as fused launches (camera + uniform; reduce + iterate + finish + slot assignment; compaction +
    filter_points) — default — or as the separate launches (tests compare both: same bits)."""
    _FUSED_SAMPLER[0] = bool(on)


class LocalTerms:
    """One step's feature-consistency term as spf_local_forward leaves it: per ray d_surface [R], lfirst [R] (-1: no crossing), lsum [R],
    lcoef [R,2], and lscale [1] — written by the loss backward (FusedLoss), read by the compositing backward (Render), which is how the term's
    gradient reaches the SDF rows without a launch of its own.  `desc`: feat_utils.LocalDesc (device descriptor of the view)."""

    def __init__(self, desc, R, dev):
        self.desc = desc
        self.d_surface = torch.empty((R,), dtype=torch.float32, device=dev)
        self.lfirst = torch.empty((R,), dtype=torch.int32, device=dev)
        self.lsum = torch.empty((R,), dtype=torch.float32, device=dev)
        self.lcoef = torch.empty((R, 2), dtype=torch.float32, device=dev)
        self.lscale = torch.empty((1,), dtype=torch.float32, device=dev)
        self.scaled = False
        self._args = None

    def args(self):
        if self._args is None:
            self._args = _lib.LocalTermsArgs(self.lsum.data_ptr(), self.lfirst.data_ptr(), self.desc.buf.data_ptr(), self.lscale.data_ptr())
        return ctypes.byref(self._args)

    def count(self):
        """source views x rays with a crossing: the denominator of the reference's mean (feat_utils.py:437), device float []."""
        n_src = self.desc.buf[_lib.LocalDesc.n_src.offset: _lib.LocalDesc.n_src.offset + 4].view(torch.int32)[0]       # read on the device: a replayed
        return (self.lfirst >= 0).sum().float() * n_src.float()                                                      # graph serves every view


def local_forward(desc, sdf, z, cam_loc, ray_dirs) -> LocalTerms:
    """find_surface_points (pointneus_disent.py:586-612) + the surface points (:744-749) + get_local_loss (feat_utils.py:377-451) for the
    rays of one view: ONE launch over the dense [R,SR] SDF / depth rows (values only; see LocalTerms for the gradient's route)."""
    R, SR = sdf.shape
    dev = sdf.device
    lt = LocalTerms(desc, R, dev)
    with torch.cuda.device(dev), _prof.span("local_fwd", rays=R, slots=SR):
        _lib.check(_lib.lib().spf_local_forward(_lib.ptr(desc.buf), _lib.ptr(sdf.detach().contiguous()), _lib.ptr(z.contiguous()),
                                                _lib.ptr(cam_loc.detach().contiguous()), _lib.ptr(ray_dirs.detach().contiguous()), R, SR,
                                                _lib.ptr(lt.d_surface), _lib.ptr(lt.lfirst), _lib.ptr(lt.lsum), _lib.ptr(lt.lcoef), _lib.stream_ptr()),
                   "spf_local_forward")
    return lt


class LocalLoss(torch.autograd.Function):
    """(sum, count, d_surface [R], hit bool [R]) of the feature-consistency term, differentiable w.r.t. sdf [R,SR] through `sum` — the
    stand-alone form (default training mode, tests): the local loss is sum / max(count, 1).  The fused optimisation step does not use it
    (LocalTerms travels through FusedLoss and Render instead: no reduction, scaling or backward launches of its own)."""

    @staticmethod
    def forward(ctx, sdf, z, cam_loc, ray_dirs, desc):
        lt = local_forward(desc, sdf, z, cam_loc, ray_dirs)
        ctx.lt, ctx.shape = lt, sdf.shape
        hit, cnt = lt.lfirst >= 0, lt.count()
        ctx.mark_non_differentiable(hit, cnt, lt.d_surface)
        return lt.lsum.sum(), cnt, lt.d_surface, hit

    @staticmethod
    def backward(ctx, g_sum, _g_cnt, _g_d, _g_hit=None):
        lt = ctx.lt
        R, SR = ctx.shape
        g_sdf = torch.empty((R, SR), dtype=torch.float32, device=lt.lsum.device)
        with torch.cuda.device(g_sdf.device):
            _lib.check(_lib.lib().spf_local_backward(_lib.ptr(lt.lfirst), _lib.ptr(lt.lcoef), _lib.ptr(g_sum.detach().reshape(1).contiguous().float()), R, SR,
                                                     _lib.ptr(g_sdf), _lib.stream_ptr()), "spf_local_backward")
        return g_sdf, None, None, None, None


_loss_ws = {}


_DEFER_LOSS = [True]


def set_loss_finalize_deferred(on=True):
    """FusedLoss: form the loss terms inside the backward launch (default; 2 launches per step) or in a finalize launch of the forward (3)."""
    _DEFER_LOSS[0] = bool(on)


class FusedLoss(_GradModeFunction):
    """(total [], terms [8]) = spf_loss_forward(...); differentiable w.r.t. rgb, acc, psdf and tv (spf_loss_backward).
    terms = {loss, rgb, eikonal, tv, mask, local, pseudo, pseudo count} (values only).
    When a backward can follow (grad mode on, a differentiable input), the forward only launches the partial sums and the BACKWARD launch forms
    the terms on its way (spf_loss_backward_finalize): `total` / `terms` then hold their values once the backward has run — an optimisation step
    always runs it; a caller that wants the loss without a backward calls this under torch.no_grad()."""

    @staticmethod
    def forward(ctx, rgb, acc, psdf, tv, grad, slot_valid, n_points, pvalid, ray_valid, rgb_gt, mask_gt, mask_stride, weights, denom, allow_defer=True,
                tv_ctx=None, local=None):
        """allow_defer=False: `total` is read by further forward ops (the feature-consistency term is added to it): finalize in the forward.
        tv_ctx = (geometry latent parameter, TVGraph): `tv` are per-point VALUES formed outside autograd (the step's first launch); their
        backward is this function's job — it rides in the loss backward launch and adds straight into the parameter's gradient sink.
        local (LocalTerms): the feature-consistency term joins the partial sums (terms[5], weighted by weights.local in the total); its
        gradient does not pass through autograd: the backward launch leaves local.lscale, which Render's backward (same LocalTerms) applies."""
        dev = rgb.device
        R = rgb.shape[0]
        rgb_c, acc_c = rgb.detach().contiguous(), acc.detach().reshape(R).contiguous()
        psdf_c = None if psdf is None else psdf.detach().reshape(R).contiguous()
        n_tv = 0 if (tv is None or tv.dim() == 0 or tv.numel() == 1) else tv.numel()       # per-point array (TVLoss reduce=False) or the mean itself
        tv_c = None if tv is None else (tv.detach().contiguous() if n_tv else tv.detach().reshape(1))
        key = _scratch_key(dev)
        if key not in _loss_ws:
            _loss_ws[key] = torch.empty((int(_lib.lib().spf_loss_workspace_floats()),), dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        terms = torch.empty((8,), dtype=torch.float32, device=dev)
        den = torch.empty((8,), dtype=torch.float32, device=dev)
        rows = 0 if grad is None else grad.shape[0]
        if local is not None and denom is not None and denom.numel() < 4:
            raise ValueError("FusedLoss(local=...): denom needs the global local count as its fourth entry (dist.fused_counts)")
        defer = bool(allow_defer) and _DEFER_LOSS[0] and _GradModeFunction._outer_grad_mode and any(ctx.needs_input_grad[:4])
        # deferred: the partial sums must survive until THIS call's backward — a buffer of their own (2048 floats; a second forward under the
        # same owner before the first backward would otherwise overwrite them: round-5 advisor finding), not the per-owner scratch
        ws = torch.empty_like(_loss_ws[key]) if defer else _loss_ws[key]
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().spf_loss_forward(_lib.ptr(rgb_c), _lib.ptr(rgb_gt), _lib.ptr(acc_c), _lib.ptr(mask_gt), mask_stride,
                                                   _lib.ptr(grad), _lib.ptr(slot_valid), rows, _lib.ptr(n_points), _lib.ptr(psdf_c),
                                                   _lib.ptr(pvalid), _lib.ptr(ray_valid), _lib.ptr(tv_c), n_tv, _lib.ptr(denom), R, weights,
                                                   _lib.ptr(ws), None if defer else _lib.ptr(total), None if defer else _lib.ptr(terms),
                                                   None if defer else _lib.ptr(den), None if local is None else local.args(), _lib.stream_ptr()),
                       "spf_loss_forward")
        ctx.save_for_backward(rgb_c, acc_c, psdf_c, rgb_gt, mask_gt, pvalid, ray_valid, den)
        ctx.misc = (mask_stride, weights, acc.shape, None if psdf is None else psdf.shape, tv is not None, n_tv)
        ctx.fin = (ws, rows, n_points, tv_c, denom, total, terms) if defer else None
        ctx.local = local
        ctx.tv_ctx = None
        if tv_ctx is not None:
            sink = _sink(tv_ctx[0])
            if sink is None or n_tv == 0:
                raise RuntimeError("FusedLoss(tv_ctx): needs the per-point TV form and a gradient sink on the latent table (ops.set_grad_sinks)")
            ctx.tv_ctx = (tv_ctx[0].detach(), tv_ctx[1], sink)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)
        return total, terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        rgb, acc, psdf, rgb_gt, mask_gt, pvalid, ray_valid, den = ctx.saved_tensors
        mask_stride, weights, acc_shape, psdf_shape, has_tv, n_tv = ctx.misc
        R, dev = rgb.shape[0], rgb.device
        if g_total is None:
            return (None,) * 17
        lt = ctx.local
        largs = None if lt is None else lt.args()
        g = g_total.detach().reshape(1).contiguous()
        g_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
        g_acc = torch.empty((R,), dtype=torch.float32, device=dev)
        g_psdf = None if psdf is None else torch.empty((R,), dtype=torch.float32, device=dev)
        g_tv = torch.empty((1,), dtype=torch.float32, device=dev) if has_tv else None
        tvc = ctx.tv_ctx
        with torch.cuda.device(dev):
            if ctx.fin is not None:
                ws, rows, n_points, tv_c, denom, total, terms = ctx.fin
                feat, graph, sink = tvc if tvc is not None else (None, None, None)
                _lib.check(_lib.lib().spf_loss_backward_finalize(_lib.ptr(g), weights, _lib.ptr(rgb), _lib.ptr(rgb_gt), _lib.ptr(acc), _lib.ptr(mask_gt), mask_stride,
                                                                 _lib.ptr(psdf), _lib.ptr(pvalid), _lib.ptr(ray_valid), R, _lib.ptr(g_rgb), _lib.ptr(g_acc),
                                                                 _lib.ptr(g_psdf), _lib.ptr(g_tv), n_tv, _lib.ptr(ws), rows, _lib.ptr(n_points), _lib.ptr(tv_c),
                                                                 _lib.ptr(denom), _lib.ptr(total), _lib.ptr(terms), _lib.ptr(den), _lib.ptr(feat),
                                                                 _lib.ptr(None if graph is None else graph.nbr), _lib.ptr(None if graph is None else graph.w),
                                                                 _lib.ptr(None if graph is None else graph.norm), 1 if graph is None else graph.nbr.shape[1],
                                                                 _lib.ptr(sink), largs, _lib.stream_ptr()), "spf_loss_backward_finalize")
            else:
                _lib.check(_lib.lib().spf_loss_backward(_lib.ptr(g), _lib.ptr(den), weights, _lib.ptr(rgb), _lib.ptr(rgb_gt), _lib.ptr(acc),
                                                        _lib.ptr(mask_gt), mask_stride, _lib.ptr(psdf), _lib.ptr(pvalid), _lib.ptr(ray_valid), R,
                                                        _lib.ptr(g_rgb), _lib.ptr(g_acc), _lib.ptr(g_psdf), _lib.ptr(g_tv), n_tv, largs, _lib.stream_ptr()),
                           "spf_loss_backward")
                if tvc is not None:          # (finalize was not deferred: the TV backward as its own launch, one gradient value for all points)
                    feat, graph, sink = tvc
                    _lib.check(_lib.lib().spf_tv_backward(_lib.ptr(feat), _lib.ptr(graph.nbr), _lib.ptr(graph.w), _lib.ptr(graph.norm), _lib.ptr(g_tv), 0, 1.0,
                                                          feat.shape[0], graph.nbr.shape[1], _lib.ptr(sink), None, _lib.stream_ptr()), "spf_tv_backward")
        if lt is not None:
            lt.scaled = True                  # lscale is on its way: Render's backward (later in autograd's order) may read it
        if tvc is not None:
            g_tv = None                       # the latent gradient has been accumulated: nothing for autograd
        g_tv_out = None if g_tv is None else (g_tv.expand(n_tv) if n_tv else g_tv.reshape(()))      # per-point array: one value, stride 0
        return (g_rgb, g_acc.view(acc_shape), None if g_psdf is None else g_psdf.view(psdf_shape), g_tv_out,
                None, None, None, None, None, None, None, None, None, None, None, None, None)
