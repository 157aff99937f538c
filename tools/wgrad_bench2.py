"""spf_wgrad (device row count) vs the split-K batched library GEMM."""
import sys
import torch
sys.path.insert(0, ".")
from tools.microbench import timeit  # noqa: E402
from spurfies_amd import ops  # noqa: E402

rows = 389000
G = torch.randn((rows + 64, 256), device="cuda")
for C in (256, 104, 21):
    A = torch.randn((rows + 64, 24 if C == 21 else C), device="cuda")
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    ref = G[:rows].t() @ A[:rows, :C]
    got = ops.wgrad(G, A, n, C=C)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    t_k = timeit(lambda: ops.wgrad(G, A, n, C=C))
    t_l = timeit(lambda: ops._wgrad(G[:388992], A[:388992]))
    print(f"C={C:3d} spf_wgrad {t_k:.3f} ms ({2*rows*256*C/t_k/1e9:.1f} TF)  library split-K {t_l:.3f} ms  rel err {err:.2e}")
