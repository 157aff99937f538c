"""How should the colour wgrad GEMMs dW = G^T A ([256 x rows] @ [rows x 256], rows ~ 4.4e5) be issued?"""
import sys
import torch
sys.path.insert(0, ".")
from tools.microbench import timeit  # noqa: E402

rows = 64 * 6900
G = torch.randn((rows, 256), device="cuda")
A = torch.randn((rows, 256), device="cuda")
ref = G.t() @ A
print("mm           ms", timeit(lambda: G.t() @ A))
for S in (4, 8, 16, 32, 64):
    Gb, Ab = G.view(S, rows // S, 256), A.view(S, rows // S, 256)
    f = lambda: torch.bmm(Gb.transpose(1, 2), Ab).sum(0)
    err = (f() - ref).abs().max().item() / ref.abs().max().item()
    print(f"bmm split {S:3d} ms", timeit(f), "rel err", err)
print("sum(0)       ms", timeit(lambda: G.sum(0)))
ones = torch.ones((1, rows), device="cuda")
print("ones@G       ms", timeit(lambda: ones @ G))
