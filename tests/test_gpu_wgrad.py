"""GPU: spf_wgrad (dW = G^T A with a device-side row count).  The 'split' family forms every fp32 product from pieces on the
16-bit matrix pipe — three bf16 pieces per operand (six piece products), or H2: two fp16 pieces per operand (three piece products)
with G block-scaled per wave and 32 columns (csrc/wgrad.hip).  This file holds both to the accuracy of the fp32-MFMA kernel against
a float64 reference — on well-scaled data, on data spanning many orders of magnitude, and on sums that cancel; every test runs once
per piece form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["bf16x3", "h2"])
def pieces(request):
    from spurfies_amd import ops

    prev = ops.set_h2(wgrad=(request.param == "h2"))
    yield request.param
    ops.set_h2(**prev)


def _err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def _both(G, A, n, C):
    from spurfies_amd import ops

    res = {}
    for mode in ("f32", "split"):
        ops.set_wgrad_mode(mode)
        try:
            res[mode] = ops.wgrad(G, A, n, C=C).clone()
        finally:
            ops.set_wgrad_mode("split")
    return res


@pytest.mark.parametrize("C", [256, 104])
@pytest.mark.parametrize("rows", [70001, 1000])
def test_split_products_match_fp32_accuracy(C, rows):
    g = torch.Generator().manual_seed(C + rows)
    G = torch.randn((rows + 40, 256), generator=g).cuda()
    A = torch.randn((rows + 40, C), generator=g).cuda()
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    ref = G[:rows].double().t() @ A[:rows].double()
    res = _both(G, A, n, C)
    e32, esp = _err(res["f32"], ref), _err(res["split"], ref)
    assert esp < 2e-6 + 2.0 * e32, (esp, e32)          # same accuracy class as the fp32 kernel (both are summation-order noise)
    assert esp < 1e-5


def test_split_products_wide_dynamic_range_and_cancellation(pieces):
    g = torch.Generator().manual_seed(7)
    rows = 20000
    scale = 10.0 ** (torch.rand((rows, 1), generator=g) * 8.0 - 4.0)                   # rows scaled over 8 orders of magnitude
    G = (torch.randn((rows, 256), generator=g) * scale).cuda()
    if pieces == "h2":
        # block floating point: a term is exact relative to the LARGEST terms of its 32-column block, so the operand to span the orders of
        # magnitude is G (as a step's does: an RBF weight and a compositing weight per row) against activations of order one — incl. entries
        # above 2^13 arriving late (the scale shrinks under way) and a block of columns that is 1e-6 of the others
        G[-200:] *= 50.0
        G[:, 32:64] *= 1e-6
        A = torch.randn((rows, 256), generator=g).cuda()
        A[:, :16] *= 1e-3                                                              # a few nearly dead units
        A[:, 16:24] *= 300.0
    else:
        A = (torch.randn((rows, 256), generator=g) / scale).cuda()                     # anti-correlated: every row's product is of order one
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    ref = G.double().t() @ A.double()
    res = _both(G, A, n, 256)
    assert _err(res["split"], ref) < 2e-6 + 2.0 * _err(res["f32"], ref)
    for blk in (slice(32, 64), slice(64, 96)):                                         # per 32-column block of G (= rows of dW), tiny or not
        assert _err(res["split"][blk], ref[blk]) < 2e-6 + 2.0 * _err(res["f32"][blk], ref[blk])
    if pieces == "h2":
        for cols in (slice(0, 16), slice(16, 24)):                                     # per column group of A
            assert _err(res["split"][:, cols], ref[:, cols]) < 4e-6 + 2.0 * _err(res["f32"][:, cols], ref[:, cols])
    # exact cancellation: every row appears twice with opposite sign -> the sum is exactly representable (0), whatever the order
    G2 = torch.cat([G[:5000], -G[:5000]]).contiguous()
    A2 = torch.cat([A[:5000], A[:5000]]).contiguous()
    n2 = torch.tensor([10000], dtype=torch.int32, device="cuda")
    res2 = _both(G2, A2, n2, 256)
    big = float((G[:5000].double().abs().t() @ A[:5000].double().abs()).max())
    assert float(res2["split"].abs().max()) <= 2e-6 * big and float(res2["f32"].abs().max()) <= 2e-6 * big
    # single rows reproduce the fp32 product of two floats to fp32 rounding: a (x) b over one row is one product per entry
    n1 = torch.tensor([1], dtype=torch.int32, device="cuda")
    r1 = _both(G[:64].contiguous(), A[:64].contiguous(), n1, 256)
    exact = torch.outer(G[0].double(), A[0].double())
    if pieces == "h2":      # ... to a few fp32 roundings of the block's largest product (entries far below it keep that absolute accuracy)
        blockmax = exact.abs().view(8, 32, 256).amax(dim=(1, 2), keepdim=True).expand(8, 32, 256).reshape(256, 256)
        assert float(((r1["split"].double() - exact).abs() / blockmax).max()) < 6e-7
    else:
        np.testing.assert_allclose(r1["split"].double().cpu().numpy(), exact.cpu().numpy(), rtol=2.5e-7, atol=0)


@pytest.mark.parametrize("C", [256, 104, 16])
def test_bias_gradient_rides_along(C):
    """dbias += column sums of G[:rows] (the bias gradient of the layer whose weight gradient is being formed), both modes."""
    from spurfies_amd import ops

    g = torch.Generator().manual_seed(C)
    rows = 33333
    G = torch.randn((rows + 100, 256), generator=g).cuda()
    A = torch.randn((rows + 100, C), generator=g).cuda()
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    ref = G[:rows].double().sum(0)
    for mode in ("f32", "split"):
        ops.set_wgrad_mode(mode)
        try:
            db = torch.full((256,), 2.0, device="cuda")
            dw = ops.wgrad(G, A, n, C=C, dbias=db)
        finally:
            ops.set_wgrad_mode("split")
        np.testing.assert_allclose((db.double() - 2.0).cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5 * float(G[:rows].abs().sum(0).max()))
        assert _err(dw, G[:rows].double().t() @ A[:rows].double()) < 1e-5


@pytest.mark.parametrize("rows", [55000, 700])
def test_three_problems_side_by_side_equal_three_calls(rows):
    """spf_wgrad_batched: three [rows,256]^T x [rows,256] GEMMs in one pair of launches (strided output, bias sums) against the
    single-problem entry point, both arithmetic modes."""
    from spurfies_amd import ops

    g = torch.Generator().manual_seed(rows)
    Gs = [torch.randn((rows + 64, 256), generator=g).cuda() for _ in range(3)]
    As = [torch.randn((rows + 64, 256), generator=g).cuda() for _ in range(3)]
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    for mode in ("split", "f32"):
        ops.set_wgrad_mode(mode)
        try:
            wide = torch.zeros((256, 277), device="cuda")                        # problem 1 writes a column block of a wider matrix
            outs = [torch.ones((256, 256), device="cuda"), wide[:, 21:], torch.zeros((256, 256), device="cuda")]
            dbs = [torch.zeros(256, device="cuda"), None, torch.full((256,), 3.0, device="cuda")]
            ops.wgrad_batched([(Gs[q], As[q], outs[q], dbs[q]) for q in range(3)], n)
            for q in range(3):
                ref = Gs[q][:rows].double().t() @ As[q][:rows].double()
                base = 1.0 if q == 0 else 0.0
                assert _err(outs[q] - base, ref) < 1e-5, (mode, q)
            assert float(wide[:, :21].abs().max()) == 0.0
            np.testing.assert_allclose(dbs[0].double().cpu().numpy(), Gs[0][:rows].double().sum(0).cpu().numpy(), rtol=0,
                                       atol=2e-5 * float(Gs[0][:rows].abs().sum(0).max()))
            np.testing.assert_allclose((dbs[2].double() - 3.0).cpu().numpy(), Gs[2][:rows].double().sum(0).cpu().numpy(), rtol=0,
                                       atol=2e-5 * float(Gs[2][:rows].abs().sum(0).max()))
        finally:
            ops.set_wgrad_mode("split")


def _to_tiles(M, rows_per_block=16):
    """[rows, 256] (rows a multiple of the block) -> the K-major image [block][256 features][16 | 64 rows] the colour kernels write."""
    t = M.shape[0] // rows_per_block
    return M.view(t, rows_per_block, 256).transpose(1, 2).contiguous().view(-1, 256)


@pytest.mark.parametrize("rows", [70001, 1000, 64, 5])
@pytest.mark.parametrize("layout", [1, 2, 3, 4, 6])
def test_tiled_operands_match_row_major(rows, layout):
    """SPF_WGRAD_G_TILES / SPF_WGRAD_A_TILES: the same GEMM with one or both operands stored as K-major tiles (what the colour
    trunk's epilogues write) — same accuracy against float64 as the row-major call, incl. a row count inside the last tile."""
    from spurfies_amd import ops

    g = torch.Generator().manual_seed(rows + layout)
    alloc = (rows + 63) // 64 * 64
    G = torch.randn((alloc, 256), generator=g).cuda()
    A = torch.randn((alloc, 256), generator=g).cuda()
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    ref = G[:rows].double().t() @ A[:rows].double()
    db_ref = G[:rows].double().sum(0)
    Gx = _to_tiles(G) if layout & 1 else (_to_tiles(G, 64) if layout & 4 else G)
    Ax = _to_tiles(A) if layout & 2 else A
    db = torch.zeros(256, device="cuda")
    got = ops.wgrad(Gx, Ax, n, dbias=db, layout=layout)
    base = ops.wgrad(G, A, n)
    assert _err(got, ref) < 2e-6 + 2.0 * _err(base, ref)
    np.testing.assert_allclose(db.double().cpu().numpy(), db_ref.cpu().numpy(), rtol=1e-5, atol=2e-5 * rows ** 0.5 + 1e-5)     # fp32 sums of `rows` N(0,1) terms, atomics in any order
    if layout in (1, 4):     # the 104-column operand of F_color's first layer stays row-major next to a blocked / tiled G
        A104 = A[:, :104].contiguous()
        got4 = ops.wgrad(Gx, A104, n, layout=layout)
        assert _err(got4, G[:rows].double().t() @ A104[:rows].double()) < 1e-5


@pytest.mark.parametrize("rows", [391000, 49000, 700, 64, 5])
def test_colour_trunk_problems_side_by_side_equal_three_calls(rows):
    """Round 4: the colour trunk's three weight-gradient GEMMs in ONE launch (spf_wgrad_batched with per-problem operand forms): layer 0 =
    G in 64-row tiles x row-major 104-column A with the column rotation into the reference order, layers 2 / 4 = G in 64-row tiles x A in
    16-row blocks; against float64 and against the single-problem entry point; bias sums ride along; row counts inside the last tile."""
    from spurfies_amd import ops

    g = torch.Generator().manual_seed(rows)
    alloc = (rows + 63) // 64 * 64
    Gs = [torch.randn((alloc, 256), generator=g).cuda() for _ in range(3)]
    A0 = torch.randn((alloc, 104), generator=g).cuda()
    As = [A0] + [torch.randn((alloc, 256), generator=g).cuda() for _ in range(2)]
    n = torch.tensor([rows], dtype=torch.int32, device="cuda")
    G64 = [_to_tiles(G_, 64) for G_ in Gs]
    A16 = [A0, _to_tiles(As[1]), _to_tiles(As[2])]
    outs = [torch.zeros((256, 103), device="cuda"), torch.ones((256, 256), device="cuda"), torch.zeros((256, 256), device="cuda")]
    dbs = [torch.zeros(256, device="cuda") for _ in range(3)]
    ops.wgrad_batched([(G64[0], A16[0], outs[0], dbs[0], 104, 4, 39, 103), (G64[1], A16[1], outs[1], dbs[1], 256, 4 | 2, 0, 0),
                       (G64[2], A16[2], outs[2], dbs[2], 256, 4 | 2, 0, 0)], n)
    # single-problem calls on the same operands
    one = [ops.wgrad(G64[0], A16[0], n, out=torch.zeros((256, 103), device="cuda"), layout=4, col_rot=39, col_mod=103),
           ops.wgrad(G64[1], A16[1], n, layout=6), ops.wgrad(G64[2], A16[2], n, layout=6)]
    ref0 = Gs[0][:rows].double().t() @ A0[:rows].double()                 # product column i lands in output column (i + 39) mod 103; column 103 is padding
    ref0 = torch.cat([ref0[:, 64:103], ref0[:, :64]], 1)
    refs = [ref0, Gs[1][:rows].double().t() @ As[1][:rows].double(), Gs[2][:rows].double().t() @ As[2][:rows].double()]
    for q in range(3):
        base = 1.0 if q == 1 else 0.0
        assert _err(outs[q] - base, refs[q]) < 2e-6 + 2.0 * _err(one[q], refs[q]), q
        np.testing.assert_allclose(dbs[q].double().cpu().numpy(), Gs[q][:rows].double().sum(0).cpu().numpy(), rtol=1e-5, atol=2e-5 * rows ** 0.5 + 1e-5)


def test_batched_entry_rejects_operand_forms_it_does_not_hold():
    from spurfies_amd import _lib, ops

    G = torch.zeros((128, 256), device="cuda")
    n = torch.tensor([128], dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.SpurfiesHipError, match="is not one of"):
        ops.wgrad_batched([(G, G, torch.zeros((256, 256), device="cuda"), None, 256, 1, 0, 0), (G, G, torch.zeros((256, 256), device="cuda"), None)], n)


@pytest.mark.parametrize("rows_trunk,rows_head", [(49000, 7100), (391000, 56000), (700, 90), (64, 5)])
def test_six_problems_with_their_own_row_counts_in_one_launch(rows_trunk, rows_head):
    """Round 5 (ABI 5): the head stage's three GEMMs (K = valid points) ride in the colour trunk's batched launch (K = pairs) — six problems,
    two row counts, three operand forms, one pair of launches; every product against float64 and no worse than the single-problem calls."""
    from spurfies_amd import ops

    g = torch.Generator().manual_seed(rows_trunk)
    at, ah = (rows_trunk + 63) // 64 * 64, (rows_head + 63) // 64 * 64 + 128        # (the head's buffers are larger than its row count: worst-case allocation)
    Gt = [torch.randn((at, 256), generator=g).cuda() for _ in range(3)]
    A0 = torch.randn((at, 104), generator=g).cuda()
    At = [A0] + [torch.randn((at, 256), generator=g).cuda() for _ in range(2)]
    Gh = [torch.randn((ah, 256), generator=g).cuda() for _ in range(3)]
    Ah = [torch.randn((ah, 256), generator=g).cuda() for _ in range(3)]
    An = torch.randn((ah, 24), generator=g).cuda()
    An[:, 21:] = 0.0
    nt = torch.tensor([rows_trunk], dtype=torch.int32, device="cuda")
    nh = torch.tensor([rows_head], dtype=torch.int32, device="cuda")
    G64 = [_to_tiles(G_, 64) for G_ in Gt]
    A16 = [A0, _to_tiles(At[1]), _to_tiles(At[2])]
    outs_t = [torch.zeros((256, 103), device="cuda"), torch.zeros((256, 256), device="cuda"), torch.zeros((256, 256), device="cuda")]
    wide = torch.zeros((256, 277), device="cuda")
    outs_h = [torch.zeros((256, 256), device="cuda"), wide[:, 21:], torch.zeros((256, 256), device="cuda")]
    dbs = [torch.zeros(256, device="cuda") for _ in range(6)]
    ops.wgrad_batched([(G64[0], A16[0], outs_t[0], dbs[0], 104, 4, 39, 103), (G64[1], A16[1], outs_t[1], dbs[1], 256, 6, 0, 0),
                       (G64[2], A16[2], outs_t[2], dbs[2], 256, 6, 0, 0)] +
                      [(Gh[q], Ah[q], outs_h[q], dbs[3 + q], 256, 0, 0, 0, nh, ah) for q in range(3)] +
                      [(Gh[1], An, wide, None, 24, 0, 0, 21, nh, ah)], nt)            # the seventh: R.0's 21 view-encoding columns (row-major, narrow)
    ref0 = Gt[0][:rows_trunk].double().t() @ A0[:rows_trunk].double()
    ref0 = torch.cat([ref0[:, 64:103], ref0[:, :64]], 1)
    refs = [ref0] + [Gt[q][:rows_trunk].double().t() @ At[q][:rows_trunk].double() for q in (1, 2)] + [Gh[q][:rows_head].double().t() @ Ah[q][:rows_head].double() for q in range(3)]
    one = [ops.wgrad(G64[0], A16[0], nt, out=torch.zeros((256, 103), device="cuda"), layout=4, col_rot=39, col_mod=103), ops.wgrad(G64[1], A16[1], nt, layout=6),
           ops.wgrad(G64[2], A16[2], nt, layout=6)] + [ops.wgrad(Gh[q], Ah[q], nh) for q in range(3)]
    for q, (got, ref) in enumerate(zip(outs_t + outs_h, refs)):
        assert _err(got, ref) < 2e-6 + 2.0 * _err(one[q], ref), q
        rows = rows_trunk if q < 3 else rows_head
        Gq = (Gt + Gh)[q]
        np.testing.assert_allclose(dbs[q].double().cpu().numpy(), Gq[:rows].double().sum(0).cpu().numpy(), rtol=1e-5, atol=2e-5 * rows ** 0.5 + 1e-5, err_msg=str(q))
    refn = Gh[1][:rows_head].double().t() @ An[:rows_head, :21].double()
    assert _err(wide[:, :21], refn) < 1e-5
