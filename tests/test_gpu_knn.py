"""GPU parity: HIP voxel-grid kNN (through the C ABI) vs the frozen-spec oracle — bit exact."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def _grid(ranges=(-1, -1, -1, 1, 1, 1)):
    from spurfies_amd.torch_knnquery import VoxelGrid

    return VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, ranges)


def test_knn_matches_golden_spec():
    fx = load_golden("knn_spec.npz")
    g = _grid()
    pts = torch.from_numpy(fx["in.pts"]).cuda()
    g.set_pointset(pts.unsqueeze(0), torch.full((1,), len(pts), dtype=torch.int32, device="cuda"))
    info = g.info()
    assert tuple(info["dims"]) == tuple(int(v) for v in fx["meta.dims"])
    assert np.array_equal(np.asarray(info["origin"], np.float32), fx["meta.origin"])
    assert info["n_in_range"] == len(pts) - 27 + 2
    for nm in ("d1", "d98", "d128"):
        x = torch.from_numpy(fx[f"{nm}.x"]).cuda()
        k, r, sr = int(fx[f"{nm}.k"]), float(fx[f"{nm}.r"]), int(fx[f"{nm}.sr"])
        d = g.query_dense(x, k, r, sr)
        assert np.array_equal(d["slot_sample"].cpu().numpy(), fx[f"{nm}.slot_sample"]), nm
        assert np.array_equal(d["pidx"].cpu().numpy(), fx[f"{nm}.pidx"]), nm
        assert np.array_equal(d["loc"].cpu().numpy(), fx[f"{nm}.loc"]), nm
        assert np.array_equal(d["ray_valid"].cpu().numpy().astype(bool), fx[f"{nm}.ray_valid"]), nm
        assert np.array_equal(d["slot_valid"].cpu().numpy().astype(bool), (fx[f"{nm}.pidx"] >= 0).any(-1)), nm
        # the op's own (compacted) return convention: dtypes and shapes of utils.py:93-113
        pidx, loc, mask = g.query(x.unsqueeze(0), k, r, sr)
        rv = fx[f"{nm}.ray_valid"]
        assert pidx.dtype == torch.int32 and loc.dtype == torch.float32 and mask.dtype == torch.int8
        assert tuple(pidx.shape) == (1, int(rv.sum()), sr, k) and tuple(mask.shape) == (1, len(rv))
        assert np.array_equal(pidx[0].cpu().numpy(), fx[f"{nm}.pidx"][rv])


def test_compaction_lists():
    from spurfies_amd import ops

    rng = np.random.default_rng(0)
    for R, SR in ((1024, 80), (131072, 1), (7, 3), (1, 1)):
        sv = (rng.uniform(size=(R, SR)) < 0.3).astype(np.uint8)
        point_slot, slot_point, n = ops.compact_points(torch.from_numpy(sv).cuda())
        want = np.nonzero(sv.reshape(-1))[0]
        assert int(n.item()) == len(want)
        assert np.array_equal(point_slot.cpu().numpy()[: len(want)], want)
        inv = np.full(R * SR, -1, np.int64)
        inv[want] = np.arange(len(want))
        assert np.array_equal(slot_point.cpu().numpy(), inv)


def test_knn_full_size_against_oracle_subset():
    """BASELINE config 5 shape (dense cloud, range +-2, 4096 rays x 98 samples): every slot of a
    random subset of rays equals the oracle; plus size-independent properties on all rays."""
    from oracle.voxel_grid import VoxelGridOracle
    from spurfies_amd import synthetic as syn

    pts, _, base = syn.make_cloud(200000, spacing=0.0125, seed=5)
    ranges = (-2, -2, -2, 2, 2, 2)
    g = _grid(ranges)
    tp = torch.from_numpy(pts).cuda()
    g.set_pointset(tp.unsqueeze(0))
    rng = np.random.default_rng(1)
    R, D = 4096, 98
    o = np.asarray([3.2, 0.4, 0.5], np.float32)
    tgt = (rng.standard_normal((R, 3)) * 0.5 * base).astype(np.float32)
    dirs = tgt - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    z = np.sort(rng.uniform(1.0, 5.5, size=(R, D)).astype(np.float32), axis=1)
    x = (o[None, None] + z[..., None] * dirs[:, None]).astype(np.float32)
    d = g.query_dense(torch.from_numpy(x).cuda(), 8, 2, 80)
    pidx = d["pidx"].cpu().numpy()
    ss = d["slot_sample"].cpu().numpy()
    # properties: slot samples strictly increasing then -1; neighbours within radius, sorted by distance, unique
    used = ss >= 0
    assert (np.diff(np.where(used, ss, 10 ** 6), axis=1) > 0)[used[:, 1:]].all()
    rr, sl, kk = np.nonzero(pidx >= 0)
    dist = np.linalg.norm(x[rr, ss[rr, sl]].astype(np.float64) - pts[pidx[rr, sl, kk]].astype(np.float64), axis=1)
    assert dist.max() <= 0.05 * (1 + 1e-5)
    first_pad = (pidx < 0).argmax(-1)
    assert ((pidx >= 0).sum(-1) == np.where((pidx < 0).any(-1), first_pad, 8)).all(), "-1 padding must be a suffix"
    # oracle on a subset of rays
    orc = VoxelGridOracle((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, ranges)
    orc.set_pointset(pts)
    assert tuple(g.info()["dims"]) == tuple(int(v) for v in orc.dims)
    sub = rng.choice(R, size=48, replace=False)
    op, ol, oss, orv = orc.query_dense(x[sub], 8, 2, 80)
    assert np.array_equal(ss[sub], oss) and np.array_equal(pidx[sub], op)
    assert np.array_equal(d["ray_valid"].cpu().numpy()[sub].astype(bool), orv)


def test_knn_edge_cases():
    g = _grid()
    far = torch.full((10, 3), 5.0, device="cuda")          # every point outside `ranges`
    g.set_pointset(far.unsqueeze(0))
    d = g.query_dense(torch.zeros((4, 3, 3), device="cuda"), 8, 2, 2)
    assert int(d["ray_valid"].sum()) == 0 and int((d["pidx"] >= 0).sum()) == 0
    one = torch.tensor([[0.1, 0.2, 0.3]], device="cuda")   # single point; query exactly on it and just outside radius
    g2 = _grid()
    g2.set_pointset(one.unsqueeze(0))
    q = torch.tensor([[[0.1, 0.2, 0.3]], [[0.1, 0.2, 0.3501]], [[0.1, 0.2, 0.3499]], [[float("nan"), 0.0, 0.0]]], device="cuda")
    d = g2.query_dense(q, 8, 2, 1)
    assert d["pidx"][:, 0, 0].tolist() == [0, -1, 0, -1]
    empty = g2.query_dense(torch.zeros((0, 5, 3), device="cuda"), 8, 2, 4)
    assert empty["pidx"].shape == (0, 4, 8)
    with pytest.raises(Exception):
        g2.query_dense(q, 9, 2, 1)  # k > SPF_KMAX is refused loudly


@pytest.mark.parametrize("compat", [("truncate",), ("layered",), ("truncate", "layered")])
def test_upstream_compat_switches_match_their_oracle_twins(compat):
    """SURVEY.md Appendix B switches (what the absent torch_knnquery source is believed to do, made deterministic): capacity limits
    per cell / per grid and the own-cell early exit — HIP against the oracle's restatement of the same rules, bit for bit, on a
    cloud dense enough that every rule bites."""
    from oracle.voxel_grid import VoxelGridOracle
    from spurfies_amd import synthetic as syn
    from spurfies_amd.torch_knnquery import VoxelGrid

    pts, _, base = syn.make_cloud(20000, spacing=0.012, seed=11)      # ~40 points per 0.075 cell
    P, max_occ = 6, 150
    args = ((0.025,) * 3, (3,) * 3, (3,) * 3, P, max_occ, (-1, -1, -1, 1, 1, 1))
    orc = VoxelGridOracle(*args, compat=compat)
    orc.set_pointset(pts)
    g = VoxelGrid(*args, compat=compat)
    g.set_pointset(torch.from_numpy(pts).cuda().unsqueeze(0))
    info = g.info()
    if "truncate" in compat:
        assert info["max_cell_points"] <= P and info["n_occupied"] <= max_occ and int(orc.occ.sum()) == info["n_occupied"]
        assert len(orc.idx_in) < len(pts) // 3, "the limits must actually drop points"
    rng = np.random.default_rng(2)
    x1 = (pts[rng.integers(0, len(pts), 3000)] + rng.normal(0, 0.01, size=(3000, 3))).astype(np.float32)
    o = np.asarray([1.6, 0.2, 0.1], np.float32)
    dirs = pts[rng.integers(0, len(pts), 64)] - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    z = np.sort(rng.uniform(0.8, 2.4, size=(64, 98)).astype(np.float32), axis=1)
    xr = (o[None, None] + z[..., None] * dirs[:, None]).astype(np.float32)
    exact = VoxelGridOracle(*args)
    exact.set_pointset(pts)
    differs = False
    for x, sr in ((x1[:, None, :], 1), (xr, 80)):
        pidx, loc, slot_sample, ray_valid = orc.query_dense(x, 8, 2, sr)
        d = g.query_dense(torch.from_numpy(x).cuda(), 8, 2, sr)
        assert np.array_equal(d["slot_sample"].cpu().numpy(), slot_sample)
        assert np.array_equal(d["pidx"].cpu().numpy(), pidx)
        assert np.array_equal(d["loc"].cpu().numpy(), loc)
        assert np.array_equal(d["ray_valid"].cpu().numpy().astype(bool), ray_valid)
        differs |= not np.array_equal(pidx, exact.query_dense(x, 8, 2, sr)[0])
    assert differs, "on this cloud the compat rules must change some neighbour lists"


def test_capacity_warning_without_truncation():
    """The exact specification keeps every point: the Python shim says so when upstream's limits would have dropped some."""
    from spurfies_amd import synthetic as syn
    from spurfies_amd.torch_knnquery import VoxelGrid

    pts, _, _ = syn.make_cloud(20000, spacing=0.012, seed=11)
    g = VoxelGrid((0.025,) * 3, (3,) * 3, (3,) * 3, 26, 20000, (-1, -1, -1, 1, 1, 1))
    with pytest.warns(UserWarning, match="max_points_per_voxel"):
        g.set_pointset(torch.from_numpy(pts).cuda().unsqueeze(0))


def test_fused_compaction_and_pair_list_equal_the_two_step_form():
    """spf_compact_pairs (two launches) against spf_compact_points + spf_build_pairs (four): identical lists, counts and fillers, from
    a single slot to the sampler pass's 131 072 slots, incl. all-invalid input."""
    from spurfies_amd import ops

    rng = np.random.default_rng(1)
    for R, SR, frac in ((1024, 80, 0.6), (131072, 1, 0.04), (7, 3, 0.5), (1, 1, 1.0), (300, 80, 0.0)):
        sv = (rng.uniform(size=(R, SR)) < frac).astype(np.uint8)
        cnt = rng.integers(1, 9, size=(R * SR,))
        nb = np.where(np.arange(8)[None, :] < cnt[:, None], rng.integers(0, 5000, size=(R * SR, 8)), -1).astype(np.int32)
        nb[sv.reshape(-1) == 0] = -1
        svt, nbt = torch.from_numpy(sv).cuda(), torch.from_numpy(nb).cuda()
        ps, sp, n = ops.compact_points(svt)
        old = ops.PairList(nbt, ps, n)
        sdf = torch.empty((R * SR,), device="cuda")
        grad = torch.empty((R * SR, 3), device="cuda")
        P, NP = old.host_counts()
        # the two-launch form (count + write) and the one-launch form (round 4: chunks publish their totals to each other), the latter three
        # times in a row: its word buffer must come back all zero
        for one_launch, reps in ((False, 1), (True, 3)):
            ops.set_compact_one_launch(one_launch)
            try:
                for _ in range(reps):
                    sdf.fill_(7.0)
                    grad.fill_(7.0)
                    new = ops.PairList.from_slots(svt, nbt, fill_sdf=sdf, fill_grad=grad)
                    assert new.host_counts() == (P, NP) and P == int(sv.sum()) and NP == int(cnt[sv.reshape(-1) == 1].sum())
                    assert torch.equal(new.point_slot[:P], ps[:P]) and torch.equal(new.slot_point, sp)
                    assert torch.equal(new.pair_off[:P + 1], old.pair_off[:P + 1]) and torch.equal(new.pair_point[:NP], old.pair_point[:NP])
                    assert bool((sdf == 1000.0).all()) and bool((grad == 0).all())
            finally:
                ops.set_compact_one_launch(True)
        for buf in ops._COMPACT_SYNC.values():
            assert not bool(buf.any()), "the one-launch compaction must leave its word buffer all zero"


def test_one_launch_compaction_is_independent_per_stream_and_at_the_largest_pass():
    """Two streams compact different inputs at the same time (MultiSceneTrainer steps scenes on alternating streams): each stream has its own
    word buffer; and the 4096-ray sampler pass (524 288 slots = 256 chunks, the largest pass of BASELINE configs[4])."""
    from spurfies_amd import ops

    rng = np.random.default_rng(5)

    def make(R, SR, frac):
        sv = (rng.uniform(size=(R, SR)) < frac).astype(np.uint8)
        cnt = rng.integers(1, 9, size=(R * SR,))
        nb = np.where(np.arange(8)[None, :] < cnt[:, None], rng.integers(0, 5000, size=(R * SR, 8)), -1).astype(np.int32)
        nb[sv.reshape(-1) == 0] = -1
        return torch.from_numpy(sv).cuda(), torch.from_numpy(nb).cuda(), int(sv.sum()), int(cnt[sv.reshape(-1) == 1].sum())

    a, b, big = make(2048, 80, 0.5), make(2048, 80, 0.3), make(4096, 128, 0.05)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for it in range(20):
        with torch.cuda.stream(s1):
            pa = ops.PairList.from_slots(a[0], a[1])
        with torch.cuda.stream(s2):
            pb = ops.PairList.from_slots(b[0], b[1])
        outs.append((pa, pb))
    torch.cuda.synchronize()
    for pa, pb in outs:
        assert pa.host_counts() == (a[2], a[3]) and pb.host_counts() == (b[2], b[3])
    assert len({k for k in ops._COMPACT_SYNC}) >= 2
    pbig = ops.PairList.from_slots(big[0].view(-1, 1), big[1])
    assert pbig.host_counts() == (big[2], big[3])
    ref = ops.PairList(big[1], *ops.compact_points(big[0].view(-1, 1))[::2])
    assert torch.equal(pbig.pair_off[:big[2] + 1], ref.pair_off[:big[2] + 1])


def test_two_captured_compactions_replayed_on_two_streams_do_not_share_words():
    """Round-4 advisor finding: under hipGraph capture every graph saw torch's capture stream, so a stream-keyed word buffer was baked into
    ALL graphs of the process; two graphs replayed at the same time then ran cp_fused_kernel on shared words (hang, or mixed lists).  Now the
    owner of a pass hands its own buffer (ops.CompactSync), and a capture without an owner takes the two-launch form.  Two graphs with
    different inputs, replayed 200 times on two streams at once, must each reproduce their own lists; the owned buffers come back zero."""
    from spurfies_amd import ops

    rng = np.random.default_rng(9)

    def make(R, SR, frac):
        sv = (rng.uniform(size=(R, SR)) < frac).astype(np.uint8)
        cnt = rng.integers(1, 9, size=(R * SR,))
        nb = np.where(np.arange(8)[None, :] < cnt[:, None], rng.integers(0, 5000, size=(R * SR, 8)), -1).astype(np.int32)
        nb[sv.reshape(-1) == 0] = -1
        return torch.from_numpy(sv).cuda(), torch.from_numpy(nb).cuda(), int(sv.sum()), int(cnt[sv.reshape(-1) == 1].sum())

    cases = [make(1024, 80, 0.5), make(1024, 80, 0.25)]
    owners = [ops.CompactSync(), ops.CompactSync()]
    refs = [ops.PairList(c[1], *ops.compact_points(c[0])[::2]) for c in cases]
    graphs, lists = [], []
    for c, own in zip(cases, owners):
        sync = own.get("main", c[0].device, c[0].numel())           # allocated (and zeroed) before the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pl = ops.PairList.from_slots(c[0], c[1], sync=sync)
        graphs.append(g)
        lists.append(pl)
    # a capture WITHOUT an owner must not bake a shared buffer in: it records the two-launch form
    n_before = len(ops._COMPACT_SYNC)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        pl3 = ops.PairList.from_slots(cases[0][0], cases[0][1])
    assert len(ops._COMPACT_SYNC) == n_before
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for it in range(200):
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
    g3.replay()
    torch.cuda.synchronize()
    for c, pl, ref in zip(cases, lists, refs):
        assert pl.host_counts() == (c[2], c[3])
        assert torch.equal(pl.pair_off[:c[2] + 1], ref.pair_off[:c[2] + 1]) and torch.equal(pl.pair_point[:c[3]], ref.pair_point[:c[3]])
    assert pl3.host_counts() == (cases[0][2], cases[0][3])
    for own in owners:
        for buf in own._bufs.values():
            assert not bool(buf.any()), "an owned word buffer must come back all zero"
