"""Where a workgroup of color_forward_kernel spends its cycles (needs a -DSPF_TIMING build:
   SPF_EXTRA_HIPCC_FLAGS=-DSPF_TIMING python -m spurfies_amd.build --force)."""
import ctypes
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
from spurfies_amd import _lib  # noqa: E402
import tools.color_bench as cb  # noqa: E402,F401  (runs the kernels)

lib = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
lib.spf_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.spf_debug_timing(buf, 1)
for _ in range(5):
    cb.fwd_train()
torch.cuda.synchronize()
lib.spf_debug_timing(buf, 1)
names = ["gather", "sync", "act0 store + GEMM1", "sync", "epilogue1", "sync", "act1 store + GEMM2", "sync", "epilogue2", "sync",
         "act2 store + GEMM3", "sync", "epilogue3 + sync", "seg reduce", "sync", "tile prologue"]
tot = sum(buf)
for n, v in zip(names, buf):
    print(f"{n:22s} {100.0 * v / tot:6.2f} %")
