"""Where a workgroup of the MLP kernels spends its cycles (needs a -DSPF_TIMING build:
   SPF_EXTRA_HIPCC_FLAGS=-DSPF_TIMING python -m spurfies_amd.build --force; add -DSPF_SOLO for one workgroup per CU in the colour trunk)."""
import ctypes
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
from spurfies_amd import _lib, ops  # noqa: E402
import tools.color_bench as cb  # noqa: E402  (sets up main-pass-shaped inputs and runs the kernels once)

lib = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()


def report(entry, fn, names):
    f = getattr(lib, entry)
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    f(buf, 1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    f(buf, 1)
    tot = sum(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    f(buf, 1)
    # thread 0 of each workgroup counts cycles from its first mark to its last: cycles per launch per workgroup / launch time
    print(entry, " ~%.2f GHz shader clock (cycle counter / wall time, %d workgroups assumed); %.0f k cycles per workgroup and launch, %.3f ms per launch"
          % (tot / 5 / 256 / (e0.elapsed_time(e1) / 5 * 1e6), 256, tot / 5 / 256 / 1e3, e0.elapsed_time(e1) / 5))
    for i, n in names.items():
        print(f"  {n:34s} {100.0 * buf[i] / tot:6.2f} %")


report("spf_debug_timing_color", cb.fwd_train,
       {15: "tile prologue", 0: "gather", 1: "sync", 2: "act0 store + GEMM1", 3: "sync", 4: "epilogue1", 5: "sync", 6: "act1 store + GEMM2",
        7: "sync", 8: "epilogue2", 9: "sync", 10: "act2 store + GEMM3", 13: "epilogue3 + weighted mean", 14: "sync"})
report("spf_debug_timing_geo",
       lambda: ops.geo_forward(cb.x, cb.pl, cb.dev["neural_pts"], cb.dev["neural_feats_geometry"], cb.packed, 45.0, with_grad=True),
       {31: "tile prologue", 0: "gather", 1: "sync", 2: "4 forward GEMMs", 3: "syncs after GEMM", 4: "4 forward epilogues", 5: "syncs after epilogue",
        6: "sdf dot", 7: "sync", 8: "Jacobian seed", 9: "sync", 10: "3 backward GEMMs", 11: "syncs after GEMM", 12: "3 backward epilogues",
        13: "syncs after epilogue", 14: "input-Jacobian GEMM + stores", 15: "sync"})
report("spf_debug_timing_color", cb.bwd_only,
       {16: "G3 = wn g_agg3 * mask -> HBM + planes", 17: "sync", 18: "2 backward GEMMs", 19: "syncs", 20: "2 backward epilogues", 21: "syncs",
        22: "next-tile lookups", 25: "latent-gradient GEMM + L store", 26: "sync + duplicate sums", 23: "sync + atomics", 24: "sync"})

# per-point head forward (training stores on), on the bench-sized point count
_P = 55600
_dev = cb.dev
_agg3 = torch.randn((_P, 256), device="cuda") * 0.1
_ws = [_dev[f"F_color.6.{w}"].clone().requires_grad_(True) for w in ("weight", "bias")] + \
      [_dev[f"R.{i}.{w}"].clone().requires_grad_(True) for i in (0, 2, 4) for w in ("weight", "bias")]
_dirs = torch.nn.functional.normalize(torch.randn((1024, 3), device="cuda"), dim=-1)
_slot = torch.arange(_P, dtype=torch.int32, device="cuda")
_n = torch.tensor([_P], dtype=torch.int32, device="cuda")
report("spf_debug_timing_rhead", lambda: ops.RHead.apply(_agg3, *_ws, _dirs, _slot, _n, 80, 81920),
       {0: "tile prologue", 1: "gather (agg3 rows, view encoding) -> planes", 2: "syncs", 3: "3 GEMMs", 4: "3 epilogues (+ 256 -> 3, sigmoid inputs)",
        5: "3 store_tile_from_planes passes", 6: "colour reduction + store"})
