import ctypes, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else ".")
os.environ["SPF_AB"] = "1"
import torch
from spurfies_amd import _lib, ops
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
f = lib.spf_debug_timing_wgrad; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
rows = 389000
G = torch.randn((rows + 64, 256), device="cuda"); A = torch.randn((rows + 64, 256), device="cuda")
n = torch.tensor([rows], dtype=torch.int32, device="cuda")
ops.wgrad(G, A, n, C=256); torch.cuda.synchronize(); f(buf, 1)
for _ in range(5): ops.wgrad(G, A, n, C=256)
torch.cuda.synchronize(); f(buf, 1)
tot = sum(buf)
for i, nm in [(0, "wait for the stage's DMA (vmcnt)"), (1, "barrier"), (2, "issue next DMAs"), (4, "split (LDS reads, VALU, plane writes)"), (5, "barrier"),
              (3, "96 MFMAs + fragment reads")]:
    print(f"{nm:40s} {100.0 * buf[i] / tot:6.2f} %   {buf[i] / (5 * 256 * 95):8.0f} ticks/stage")
