"""Where waves 0 (splits first) and 4 (multiplies first) of a weight-gradient workgroup spend the stage loop (needs a -DSPF_TIMING build)."""
import ctypes
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
from spurfies_amd import _lib, ops  # noqa: E402

rows = 391040
G = torch.randn((rows, 256), device="cuda")
A = torch.randn((rows, 256), device="cuda")
n = torch.tensor([rows], dtype=torch.int32, device="cuda")
lib = _lib.lib()
f = lib.spf_debug_timing_wgrad
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 32)()
layout = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(20):
    ops.wgrad(G, A, n, layout=layout)
torch.cuda.synchronize()
f(buf, 1)
for _ in range(50):
    ops.wgrad(G, A, n, layout=layout)
torch.cuda.synchronize()
f(buf, 1)
names = ["DMA issue", "split: rest (plane writes landed)", "48 MFMAs + B reads", "vmcnt/lgkmcnt wait", "barrier", "split: A rows read", "split: A arithmetic",
         "split: plane writes issued + G rows read", "split: G arithmetic"]
for w in (0, 1):
    st = buf[16 * w + 10]
    print("wave %d: %d stages; cycles per stage: " % (4 * w, st) + ", ".join("%s %.0f" % (names[i], buf[16 * w + i] / st) for i in range(9))
          + "; total %.0f" % (sum(buf[16 * w + i] for i in range(9)) / st))
