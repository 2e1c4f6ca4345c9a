"""GPU: the brute-force kNN producer returns scikit-learn's exact neighbours (the reference's
CPU implementation, neighbors.py:106-107,242) and squared-l2 distances."""

import numpy as np
import pytest

from tests.util import to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,k", [(1, 5), (8, 10), (40, 30)])
def test_matches_sklearn(d, k):
    from sklearn.neighbors import NearestNeighbors

    from muygpys_amd.neighbors import NN_Wrapper

    rng = np.random.default_rng(d)
    X = rng.normal(size=(3000, d))
    Q = rng.normal(size=(257, d))
    nn = NN_Wrapper(to_dev(X, torch.float64), k, chunk=100)
    idx, dist = nn.get_nns(to_dev(Q, torch.float64))
    ref = NearestNeighbors(n_neighbors=k, algorithm="brute").fit(X)
    rd, ri = ref.kneighbors(Q)
    assert np.array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_allclose(dist.cpu().numpy(), rd**2, rtol=1e-9, atol=1e-12)
    bi = rng.choice(3000, size=100, replace=False)
    bidx, bdist = nn.get_batch_nns(to_dev(bi))
    rd, ri = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(X).kneighbors(X[bi])
    assert np.array_equal(bidx.cpu().numpy(), ri[:, 1:])
    np.testing.assert_allclose(bdist.cpu().numpy(), rd[:, 1:] ** 2, rtol=1e-9, atol=1e-12)
    assert not (bidx.cpu().numpy() == bi[:, None]).any()


@pytest.mark.parametrize("scan_kind", ["bf16x3", "f32"])
@pytest.mark.parametrize("d,k,n", [(40, 30, 25000), (8, 50, 40000)])
def test_mfma_scans_match_sklearn_index_exactly(d, k, n, scan_kind):
    """Both MFMA scans (past the 4096-row dense start) against the reference's implementation --
    scikit-learn brute force (neighbors.py:106-107,242) -- on the same fp32 data: the same indices in
    the same order, except where the two candidates at a position are tied to fp32 rounding (their
    fp64 distances from the query differ by less than the fp32 rounding of a d-term sum, (2 + sqrt(d)/2) ulp)."""
    from sklearn.neighbors import NearestNeighbors

    from muygpys_amd.neighbors import NN_Wrapper

    rng = np.random.default_rng(100 * d + k)
    X = rng.normal(size=(n, d)).astype(np.float32)
    Q = rng.normal(size=(700, d)).astype(np.float32)
    nn = NN_Wrapper(to_dev(X, torch.float32), k, scan_kind=scan_kind)
    assert nn._scan_supported(to_dev(Q, torch.float32), k)
    ref = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(X)
    bi = rng.choice(n, size=700, replace=False)
    bi[:4] = [0, 4095, 4096, n - 1]
    for query, got, want in (
        ("test", nn.get_nns(to_dev(Q, torch.float32)), ref.kneighbors(Q, n_neighbors=k, return_distance=False)),
        ("batch", nn.get_batch_nns(to_dev(bi)), ref.kneighbors(X[bi], return_distance=False)[:, 1:]),
    ):
        assert int(nn.last_overflow.sum()) == 0, "the scan itself must have produced these lists"
        idx = got[0].cpu().numpy()
        pts = Q if query == "test" else X[bi]
        diff = np.argwhere(idx != want)
        for r, c in diff:  # a mismatch is only acceptable between candidates tied to fp32 rounding
            da = ((pts[r].astype(np.float64) - X[idx[r, c]].astype(np.float64)) ** 2).sum()
            db = ((pts[r].astype(np.float64) - X[want[r, c]].astype(np.float64)) ** 2).sum()
            # fp32 rounding of a d-term sum of squares: 2 ulp + sqrt(d) / 2 ulp of accumulation
            assert abs(da - db) <= (2 + 0.5 * np.sqrt(d)) * np.spacing(np.float32(max(da, db))), (query, r, c, da, db)
        assert len(diff) <= 0.002 * idx.size
        if query == "batch":
            assert not (idx == bi[:, None]).any()


def test_end_to_end_with_true_neighbours():
    """kNN producer -> fused posterior vs the oracle on the same neighbourhoods."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var
    from muygpys_amd.neighbors import NN_Wrapper
    from oracle import muygps_oracle as orc
    from tests.util import RTOL, assert_close

    rng = np.random.default_rng(5)
    X = rng.normal(size=(5000, 40))
    y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=5000)
    bi = np.arange(0, 5000, 10)
    Xd, yd, bid = to_dev(X, torch.float32), to_dev(y, torch.float32), to_dev(bi)
    ni, _ = NN_Wrapper(Xd, 30).get_batch_nns(bid)
    mean, var = posterior_mean_var(KernelSpec("matern15", "l2", 5.0, 1e-3), Xd, Xd, bid, ni, yd)
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern15", "l2", 5.0, 1e-3), X, X, bi, ni.cpu().numpy(), y)
    assert_close(mean.cpu().numpy(), m_ref, RTOL["float32"], "mean")
    assert_close(var.cpu().numpy(), v_ref, RTOL["float32"], "var")


@pytest.mark.parametrize("scan_kind", ["bf16x3", "f32"])
@pytest.mark.parametrize("d,k,n,m", [(40, 30, 30000, 3000), (8, 50, 50000, 2500), (36, 10, 20000, 1000), (64, 64, 12000, 777),
                                     (24, 30, 40000, 2000), (16, 20, 30000, 1500)])  # (32-slot packed rows)
def test_mfma_scan_matches_dense_path(d, k, n, m, scan_kind):
    """The fused MFMA scan (mgp_knn_scan_f32) and the dense matmul+topk path return the same
    neighbour distances (index sets may differ only where distances tie to fp32 rounding)."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(d * 1000 + k)
    X = torch.randn(n, d, generator=g).cuda()
    Q = torch.randn(m, d, generator=g).cuda()
    scan, dense = NN_Wrapper(X, k, scan_kind=scan_kind), NN_Wrapper(X, k, use_scan=False)
    assert scan._scan_supported(Q, k)
    for query in ("test", "batch"):
        if query == "test":
            (i1, d1), (i2, d2) = scan.get_nns(Q), dense.get_nns(Q)
        else:
            bi = torch.randperm(n, generator=g)[:m].cuda()
            bi[:5] = torch.tensor([0, 1, 4095, 4096, n - 1], device="cuda")  # both sides of the init rows
            (i1, d1), (i2, d2) = scan.get_batch_nns(bi), dense.get_batch_nns(bi)
            assert not (i1 == bi[:, None]).any()
        assert i1.dtype == torch.int64 and i1.shape == (m, k)
        torch.testing.assert_close(d1, d2, rtol=1e-5, atol=1e-6)
        assert float((i1 == i2).float().mean()) > 0.999
        assert bool((d1[:, 1:] >= d1[:, :-1]).all())
        assert int(i1.min()) >= 0 and int(i1.max()) < n
        assert all(len(set(row.tolist())) == k for row in i1[:50])


@pytest.mark.parametrize("scan_kind", ["bf16x3", "f32"])
def test_mfma_scan_survives_adversarial_row_order(scan_kind):
    """Rows sorted by distance from the queries, farthest first: every tile beats the running
    k-th best, queues overflow, and the flagged queries are recomputed on the dense path."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(3)
    X = torch.randn(20000, 8, generator=g)
    X = X[(X**2).sum(1).argsort(descending=True)].contiguous().cuda()
    Q = (0.01 * torch.randn(300, 8, generator=g)).cuda()  # near the origin
    # shuffle=False keeps the adversarial order in front of the kernel (the default storage order is
    # pseudo-random precisely so that this does not happen)
    (i1, d1), (i2, d2) = (NN_Wrapper(X, 20, scan_kind=scan_kind, shuffle=False).get_nns(Q),
                          NN_Wrapper(X, 20, use_scan=False).get_nns(Q))
    torch.testing.assert_close(d1, d2, rtol=1e-5, atol=1e-6)
    assert float((i1 == i2).float().mean()) > 0.999


def test_split_bf16_scan_is_exact_far_from_the_origin():
    """The pre-filter's margin scales with |q| |x|: a table far from the origin (offset 50, neighbour
    distances ~1) lets more near misses through but must not lose a true neighbour.  Reference: fp64
    brute force (the fp32 Gram form of the dense path itself loses ~5e-3 of absolute accuracy here)."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(9)
    X = (50.0 + torch.randn(30000, 8, generator=g)).cuda()
    Q = (50.0 + torch.randn(500, 8, generator=g)).cuda()
    i1, d1 = NN_Wrapper(X, 25, scan_kind="bf16x3").get_nns(Q)
    ref = torch.cdist(Q.double(), X.double()) ** 2
    d2, i2 = ref.topk(25, dim=1, largest=False)
    torch.testing.assert_close(d1.double(), d2, rtol=1e-4, atol=1e-5)
    assert float((i1 == i2).float().mean()) > 0.999


def test_spatially_sorted_table_stays_on_the_scan_path():
    """A table sorted along its first feature puts every query's neighbours in a few consecutive
    rows; NN_Wrapper stores scan tables in a pseudo-random row order, so the candidate queues do not
    overflow (no dense-path recomputation) and the result still refers to the caller's rows."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(12)
    X = torch.rand(200_000, 4, generator=g)
    X = X[X[:, 0].argsort()].contiguous().cuda()
    bi = torch.arange(0, 200_000, 7, device="cuda")
    nbrs = NN_Wrapper(X, 16)
    idx, dist = nbrs.get_batch_nns(bi)
    assert int(nbrs.last_overflow.sum()) == 0, "no query may have fallen back to the dense path"
    ref = torch.cdist(X[bi[:500]].double(), X.double()) ** 2
    ref[torch.arange(500), bi[:500]] = float("inf")
    rd, ri = ref.topk(16, dim=1, largest=False)
    torch.testing.assert_close(dist[:500].double(), rd, rtol=1e-4, atol=1e-7)
    assert float((idx[:500] == ri).float().mean()) > 0.999


@pytest.mark.parametrize("d,k,n,m", [(8, 50, 300000, 4000), (4, 10, 150000, 3000)])
def test_two_chain_d8_scan_returns_the_three_chain_scans_lists(d, k, n, m, monkeypatch):
    """d <= 8 runs the two-chain kernel on 24-slot rows (mgp_knn_scan_bf16x2_d8, round 5); MUYGPYS_HIP_KNN_D8=0 keeps
    the three-chain kernel on [hi(16) | lo(16)] rows.  Both pre-filters feed the same exact re-measurement: the same
    lists, index for index (enough rows for the drain interval to grow past a tile)."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(41 + d)
    X = torch.randn(n, d, generator=g).cuda()
    Q = torch.randn(m, d, generator=g).cuda()
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MUYGPYS_HIP_KNN_D8", flag)
        nn = NN_Wrapper(X, k, scan_kind="bf16x3")
        out[flag] = nn.get_nns(Q)
        assert nn._packed_train.shape[1] == (24 if flag == "1" else 32)
        assert int(nn.last_overflow.sum()) == 0
    assert torch.equal(out["1"][0], out["0"][0])
    assert torch.equal(out["1"][1], out["0"][1])
    ref = torch.cdist(Q[:500].double(), X.double()) ** 2
    d2, _ = ref.topk(k, dim=1, largest=False)
    torch.testing.assert_close(out["1"][1][:500].double(), d2, rtol=1e-5, atol=1e-6)


def test_four_row_block_scan_returns_the_same_lists(monkeypatch):
    """From 2**20 queries on, the d <= 8 scan runs four row blocks of 32 queries per wave (512 per workgroup);
    MUYGPYS_HIP_KNN_RB4_MIN=0 forces that kernel at any size: the same lists as the two-block kernel, index for index
    (a query count that is not a multiple of 512: the last workgroup is partly empty)."""
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(77)
    X = torch.randn(200000, 8, generator=g).cuda()
    Q = torch.randn(5000 + 37, 8, generator=g).cuda()
    out = {}
    for flag in ("0", str(1 << 40)):
        monkeypatch.setenv("MUYGPYS_HIP_KNN_RB4_MIN", flag)
        nn = NN_Wrapper(X, 50, scan_kind="bf16x3")
        out[flag] = nn.get_nns(Q)
        assert int(nn.last_overflow.sum()) == 0
    (i4, d4), (i2, d2) = out["0"], out[str(1 << 40)]
    assert torch.equal(d4, d2) and float((i4 == i2).float().mean()) > 0.9999
    ref = torch.cdist(Q[:300].double(), X.double()) ** 2
    torch.testing.assert_close(d4[:300].double(), ref.topk(50, dim=1, largest=False)[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,cols,k", [(1, 64, 1), (257, 1000, 30), (1000, 1920, 50), (33, 4096, 64), (5, 70, 70), (64, 129, 7)])
def test_topk_rows_kernel_selects_the_k_smallest(rows, cols, k):
    """mgp_topk_rows_f32 (round 6: the scans' initial lists, off torch.topk): per row the same SET of values as
    torch.topk(largest=False), the reported columns hold those values, no column twice -- with ties, infinities
    (the excluded self-match) and a padded last lane group."""
    from muygpys_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(rows * 7 + cols)
    x = torch.randn((rows, cols), device="cuda", generator=g)
    x[:, ::5] = x[:, ::5].round(decimals=1)           # ties
    x[torch.arange(rows, device="cuda"), torch.randint(0, cols, (rows,), device="cuda", generator=g)] = float("inf")
    vals = torch.empty((rows, k), device="cuda")
    idx = torch.empty((rows, k), device="cuda", dtype=torch.int32)
    rc = _lib.load().mgp_topk_rows_f32(_lib.ptr(x), rows, cols, x.stride(0), k, _lib.ptr(vals), _lib.ptr(idx), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    ref = x.topk(k, dim=1, largest=False).values.sort(dim=1).values
    assert torch.equal(vals.sort(dim=1).values, ref)
    assert torch.equal(x.gather(1, idx.long()), vals)
    assert all(len(set(r.tolist())) == k for r in idx.cpu())


@pytest.mark.parametrize("m,k,d", [(1, 1, 4), (1000, 30, 40), (777, 50, 8), (130, 64, 16), (65, 17, 12)])
def test_knn_finish_kernel_orders_and_maps(m, k, d):
    """mgp_knn_finish_f32: exact difference-form distances of the candidates, ascending, ties by position in the list,
    indices through the row map -- against the torch expression it replaced."""
    from muygpys_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + k)
    n = 5000
    train = torch.randn((n, d), device="cuda", generator=g)
    q = torch.randn((m, d), device="cuda", generator=g)
    cand = torch.stack([torch.randperm(n, device="cuda", generator=g)[:k] for _ in range(m)]).to(torch.int32)
    if k > 2:  # an exact tie per query: list position 1 points at a COPY of the row at position 0
        train = torch.cat([train, train[cand[:, 0].long()]]).contiguous()
        cand[:, 1] = n + torch.arange(m, device="cuda", dtype=torch.int32)
        n = n + m
    perm = torch.randperm(n, device="cuda", generator=g)
    idx = torch.empty((m, k), device="cuda", dtype=torch.int64)
    dist = torch.empty((m, k), device="cuda")
    for row_map in (None, perm):
        rc = _lib.load().mgp_knn_finish_f32(_lib.ptr(q), _lib.ptr(train), d, _lib.ptr(cand), m, k, _lib.ptr(row_map),
                                            _lib.ptr(idx), _lib.ptr(dist), _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        dd = ((q[:, None, :] - train[cand.long()]) ** 2).sum(-1)
        torch.testing.assert_close(dist, dd.sort(dim=1).values, rtol=2e-6, atol=1e-6)
        assert bool((dist[:, 1:] >= dist[:, :-1]).all())
        # the same multiset of rows per query; where distances differ clearly, the same order
        want = cand.long() if row_map is None else perm[cand.long()]
        assert torch.equal(idx.sort(dim=1).values, want.sort(dim=1).values)
        order = dd.argsort(dim=1, stable=True)
        if k > 1:
            clear = (dd.gather(1, order)[:, 1:] - dd.gather(1, order)[:, :-1]).abs().min(dim=1).values > 1e-4
            assert torch.equal(idx[clear], want.gather(1, order)[clear])
        else:
            assert torch.equal(idx, want)
        if k > 2:  # the planted tie: list positions 0 and 1 carry the same row -> position 0 first (a stable order)
            p0 = (idx == want[:, :1]).float().argmax(dim=1)
            p1 = (idx == want[:, 1:2]).float().argmax(dim=1)
            assert bool((p0 < p1).all())
