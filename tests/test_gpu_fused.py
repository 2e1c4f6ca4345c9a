"""GPU parity of the fused HIP path (mgp_posterior_*) against the oracle / golden fixtures."""

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.conftest import spec_from_meta
from tests.util import RTOL, assert_close, assert_rel_close, fp32_reference, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _kspec(meta, g, dtype):
    from muygpys_amd.fused import KernelSpec

    ls = meta["length_scale"]
    noise = to_dev(g["noise_table"], dtype) if meta.get("hetero") else float(meta["noise"])
    return KernelSpec(kernel=meta["kernel"], metric=meta["metric"], length_scale=ls, noise=noise)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("route", ["auto", "auto-prepared-tables", "generic"])
def test_fused_matches_golden(golden, dtype, route):
    """Every fixture through the dispatcher's kernel on the plain tables, through the prepared
    tables (mgp_table_pack_* + mgp_posterior_packed_*) and through the LDS workgroup kernel."""
    from muygpys_amd.fused import PackedTable, posterior_mean_var

    g, meta = golden, golden["meta"]
    td = getattr(torch, dtype)
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        # reference itself skips fp32 solves at small d (tests/gp.py:514-519): conditioning
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    packed = route == "auto-prepared-tables"
    if packed and not PackedTable.supported(meta["d"], meta["R"], meta["k"], td):
        pytest.skip("shape outside the prepared-table kernels (rows not 16-byte multiples / too many responses)")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var, yk = posterior_mean_var(
        _kspec(meta, g, td), X, X, bi, ni, y, want_ykinvy=True, info=info,
        path="generic" if route == "generic" else "auto", packed=packed,
    )
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    rtol = RTOL[dtype]
    assert_close(mean.cpu().numpy(), g["mean"], rtol, "mean")
    assert_close(var.cpu().numpy(), g["var_unscaled"], rtol, "var")
    # the variance is strictly positive: also within the stated tolerance in the pure relative sense
    # (fp32: calibrated against the reference's OWN fp32 backend on the same inputs, no absolute floor)
    var32_ref = fp32_reference(meta["name"])[1] if dtype == "float32" else None
    assert_rel_close(var.cpu().numpy(), g["var_unscaled"], rtol, "var (relative)", calibration=var32_ref)
    b, k = g["nn_idx"].shape
    sig = yk.double().sum(dim=0).cpu().numpy().reshape(-1) / (b * k)
    assert_rel_close(sig, g["sigma_sq"], rtol, "sigma_sq")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_prepared_tables_give_identical_bits(dtype):
    """The prepared-table route runs the same kernels on the same values: bitwise equal outputs,
    with a separate (response-free) query table too."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    td = getattr(torch, dtype)
    gen = torch.Generator(device="cuda").manual_seed(5)
    # (b below MUYGPYS_HIP_JIT_CACHED_MIN_BATCH: above it the two routes may be served by different specialised
    # kernels -- the build caches the fp64 shapes for prepared tables only -- whose summation orders differ)
    N, M, d, k, b = 30000, 5000, 40 if dtype == "float32" else 8, 30, 4001
    X = torch.randn(N, d, device="cuda", dtype=td, generator=gen)
    Q = torch.randn(M, d, device="cuda", dtype=td, generator=gen)
    y = torch.randn(N, device="cuda", dtype=td, generator=gen)
    bi = torch.randint(0, M, (b,), device="cuda", generator=gen)
    ni = torch.randint(0, N, (b, k), device="cuda", generator=gen)
    spec = KernelSpec("matern15", "l2", float(np.sqrt(2 * d)), 1e-3)
    ref = posterior_mean_var(spec, Q, X, bi, ni, y, want_ykinvy=True, packed=False)
    got = posterior_mean_var(spec, Q, X, bi, ni, y, want_ykinvy=True, packed=True)
    torch.cuda.synchronize()
    for a, c in zip(ref, got):
        assert torch.equal(a, c)


def test_query_table_stride_keeps_the_response_slot_inside_the_row():
    """Round-2 advisor finding: the gather reads the 16-byte slot behind the features with EVERY row,
    also of a query table packed without responses.  With d * s a multiple of 64 (d = 16, fp32) the
    old stride ended at the features, so that read ran 16 bytes past the last row of the table.  The
    stride now reserves the slot; the library refuses a stride without it."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, PackedTable, posterior_mean_var

    lib = _lib.load()
    for d, es in ((16, 4), (32, 4), (8, 8), (40, 4)):
        for R in (0, 1):
            assert lib.mgp_packed_row_bytes(d, R, es) >= d * es + 16
            assert lib.mgp_packed_row_bytes(d, R, es) % 64 == 0
    gen = torch.Generator(device="cuda").manual_seed(11)
    N, M, d, k = 4096, 1 << 14, 16, 30
    X = torch.randn(N, d, device="cuda", generator=gen)
    Q = torch.randn(M, d, device="cuda", generator=gen)
    y = torch.randn(N, device="cuda", generator=gen)
    bi = torch.arange(M - 512, M, device="cuda")  # the last rows of the query table, the very last included
    ni = torch.randint(0, N, (bi.numel(), k), device="cuda", generator=gen)
    spec = KernelSpec("matern15", "l2", float(np.sqrt(2 * d)), 1e-3)
    ref = posterior_mean_var(spec, Q, X, bi, ni, y, packed=False)
    got = posterior_mean_var(spec, Q, X, bi, ni, y, packed=True)
    torch.cuda.synchronize()
    assert PackedTable(Q).stride >= d * 4 + 16
    for a, c in zip(ref, got):
        assert torch.equal(a, c)
    # a caller-made query table without the slot is refused (MGP_EINVAL), not read out of bounds
    pn = PackedTable(X, y)
    mean = torch.empty((bi.numel(), 1), device="cuda")
    var = torch.empty((bi.numel(),), device="cuda")
    ls = torch.tensor([5.0], device="cuda")
    rc = _lib.fn("posterior_packed", torch.float32)(
        _lib.ptr(Q), d * 4, _lib.ptr(pn.data), pn.stride, d, _lib.ptr(bi), _lib.ptr(ni), bi.numel(), k, 1,
        0, 1e-3, None, spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), 1,
        _lib.ptr(mean), _lib.ptr(var), None, None, _lib.stream_ptr(),
    )
    assert rc == -1  # MGP_EINVAL


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_random_large(dtype):
    """BASELINE config-2 shape at a size the oracle finishes in seconds."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    rng = np.random.default_rng(3)
    N, d, k, b = 20000, 40, 30, 3000
    X = rng.normal(size=(N, d))
    y = np.sin(X @ rng.normal(size=d) / np.sqrt(d)) + 0.1 * rng.normal(size=N)
    bi = rng.choice(N, size=b, replace=False)
    ni = rng.integers(0, N - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])  # never the point itself
    ospec = orc.Spec("matern15", "l2", 5.0, 1e-3)
    m_ref, v_ref = orc.posterior_mean_var_chunked(ospec, X, X, bi, ni, y, chunk=500)
    td = getattr(torch, dtype)
    mean, var = posterior_mean_var(
        KernelSpec("matern15", "l2", 5.0, 1e-3), to_dev(X, td), to_dev(X, td), to_dev(bi), to_dev(ni), to_dev(y, td)
    )
    torch.cuda.synchronize()
    assert_close(mean.cpu().numpy(), m_ref, RTOL[dtype], "mean")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL[dtype], "var")


def test_non_spd_is_flagged():
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    # duplicate neighbours + zero nugget -> exactly singular K
    X = torch.randn(50, 4, device="cuda", dtype=torch.float64)
    y = torch.randn(50, device="cuda", dtype=torch.float64)
    ni = torch.tensor([[1, 1, 2, 3], [4, 5, 6, 7]], device="cuda")
    bi = torch.tensor([0, 8], device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec("rbf", "F2", 1.0, 0.0), X, X, bi, ni, y, info=info)
    torch.cuda.synchronize()
    assert int(info.item()) == 1
    assert torch.isnan(mean[0]) and torch.isnan(var[0])
    assert torch.isfinite(mean[1]) and torch.isfinite(var[1])


@pytest.mark.parametrize("mode", ["aniso", "hetero_table", "hetero_batch", "no_batch_idx", "odd_batch"])
def test_static_headline_shape_variants(mode):
    """The compile-time-shape instantiation (k=30, d=40, R=1, fp32) with the run-time options it
    still has to honour: per-feature length scales, both heteroscedastic layouts, identity batch
    indices, an odd number of neighbourhoods (half-empty last wave)."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    rng = np.random.default_rng(99)
    N, d, k, b = 3000, 40, 30, 257
    X = rng.normal(size=(N, d))
    y = np.sin(X[:, :3].sum(1)) + 0.1 * rng.normal(size=N)
    bi = rng.choice(N, size=b, replace=False)
    ni = np.stack([rng.choice(np.delete(np.arange(N), i), size=k, replace=False) for i in bi])
    ls, noise_o, noise_d = 5.0, 1e-3, 1e-3
    Q, bi_o, bi_d = X, bi, to_dev(bi)
    if mode == "aniso":
        ls = list(rng.uniform(3.0, 8.0, size=d))
    elif mode == "hetero_table":
        noise_o = 10.0 ** rng.uniform(-4, -2, size=N)
        noise_d = to_dev(noise_o, torch.float32)
    elif mode == "hetero_batch":
        table = 10.0 ** rng.uniform(-4, -2, size=N)
        noise_o = table
        noise_d = to_dev(table[ni], torch.float32)
    elif mode == "no_batch_idx":
        Q = X[bi]
        bi_o, bi_d = np.arange(b), None
    ospec = orc.Spec("matern15", "l2", np.asarray(ls) if isinstance(ls, list) else ls, noise_o)
    m_ref, v_ref = orc.posterior_mean_var(ospec, Q, X, bi_o, ni, y)
    spec = KernelSpec("matern15", "l2", ls, noise_d)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(spec, to_dev(Q, torch.float32), to_dev(X, torch.float32), bi_d, to_dev(ni),
                                   to_dev(y, torch.float32), info=info)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    assert_close(mean.cpu().numpy(), m_ref, RTOL["float32"], f"mean ({mode})")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var ({mode})")


def test_folded_elimination_pairs_every_batch_size():
    """The folded elimination (csrc/mgp_fused_wave_kernel.h, phase 4F) works on PAIRS of a workgroup's tasks:
    every small batch size -- a workgroup with one task, an unpaired last task, a half-empty last wave, fewer
    tasks than workgroups -- against the oracle, and a singular neighbourhood flagged on its own whichever half of
    whichever task of a pair it sits in (the four neighbourhoods of one elimination share every instruction)."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    rng = np.random.default_rng(4)
    N, d, k = 2000, 40, 30
    X = rng.normal(size=(N, d))
    y = np.sin(X[:, :3].sum(1))
    Xd, yd = to_dev(X, torch.float32), to_dev(y, torch.float32)
    spec, ospec = KernelSpec("matern15", "l2", 5.0, 1e-3), orc.Spec("matern15", "l2", 5.0, 1e-3)
    for b in (1, 2, 3, 4, 5, 7, 8, 9, 6143, 6144, 6145, 12289, 12291):
        bi = rng.choice(N, size=b, replace=True)
        ni = np.stack([rng.choice(np.delete(np.arange(N), i), size=k, replace=False) for i in bi])
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        mean, var, yk = posterior_mean_var(spec, Xd, Xd, to_dev(bi), to_dev(ni), yd, want_ykinvy=True, info=info)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        pick = np.unique(np.concatenate([np.arange(min(b, 40)), np.arange(max(b - 40, 0), b)]))
        m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi[pick], ni[pick], y)
        assert_close(mean.cpu().numpy()[pick], m_ref, RTOL["float32"], f"mean (b={b})")
        assert_rel_close(var.cpu().numpy()[pick], v_ref, RTOL["float32"], f"var (b={b})")
        assert torch.isfinite(yk).all()
    # singular neighbourhoods (a duplicated neighbour, zero nugget): each position flagged alone
    b = 24581
    bi = rng.choice(N, size=b, replace=True)
    ni = np.stack([rng.choice(np.delete(np.arange(N), i), size=k, replace=False) for i in bi])
    broken = np.array([0, 1, 2, 3, 12290, 12291, 12293, b - 1])
    ni[broken, 1] = ni[broken, 0]
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec("matern15", "l2", 5.0, 0.0), Xd, Xd, to_dev(bi), to_dev(ni), yd, info=info)
    torch.cuda.synchronize()
    nan_rows = torch.isnan(var).cpu().numpy()
    assert int(info.item()) == len(broken)
    assert sorted(np.flatnonzero(nan_rows)) == sorted(broken)
    assert torch.isnan(mean.reshape(b, -1)[torch.from_numpy(broken).cuda()]).all()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_rhs_columns_kernel_matches_golden(golden, dtype):
    """The right-hand-sides-as-columns kernel (mgp_fused_rhs.hip: k <= 64, R <= 16) on every
    fixture it covers, forced ahead of the row form."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import posterior_mean_var

    g, meta = golden, golden["meta"]
    if meta["k"] > 64 or meta["R"] > 16:
        pytest.skip("outside the kernel's range")
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")
    td = getattr(torch, dtype)
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var, yk = posterior_mean_var(_kspec(meta, g, td), X, X, bi, ni, y, want_ykinvy=True, info=info, path="rhs")
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    rtol = RTOL[dtype]
    assert_close(mean.cpu().numpy(), g["mean"], rtol, "mean")
    assert_close(var.cpu().numpy(), g["var_unscaled"], rtol, "var")
    b, k = g["nn_idx"].shape
    sig = yk.double().sum(dim=0).cpu().numpy().reshape(-1) / (b * k)
    assert_close(sig, g["sigma_sq"], rtol, "sigma_sq")


BACK_CASES = [
    # dtype, kernel, k, d, R, path, b -- the prediction variant of the rhs-column kernel (fused_rhs_kernel<T,16,BACK=true>:
    # 5 <= R <= 16 and no y^T K^-1 y requested): what BASELINE config 5's bench line times.  Gram form (fp32, d a
    # multiple of 8, not Matern-1/2) and difference form; k on and off the 8- and 32-step block boundaries of the
    # elimination / back-substitution; one launch long enough for every workgroup to loop over many tasks.
    ("float32", "rbf", 64, 40, 16, "auto", 66_000),
    ("float32", "rbf", 64, 40, 16, "auto", 1200),
    ("float32", "matern15", 50, 40, 8, "auto", 1200),       # k + 1 + R = 59 fits the row form: auto must still be right
    ("float32", "matern15", 50, 40, 16, "auto", 1200),      # 67 rows -> rhs columns, BACK, Gram
    ("float32", "matern25", 33, 40, 5, "rhs", 1200),
    ("float32", "matern05", 64, 6, 16, "auto", 1200),       # difference form (Matern-1/2, d = 6)
    ("float32", "matern05", 50, 6, 8, "rhs", 1200),
    ("float32", "rbf", 33, 6, 16, "rhs", 1200),
    ("float32", "matern15", 20, 8, 5, "rhs", 1200),         # k < 24: the rows 24..31 of the multiplier matrix are never written
    ("float32", "matern15", 40, 16, 8, "rhs", 1200),
    ("float32", "matern15", 48, 24, 16, "auto", 1200),      # k = 48: the advisor's reachable-on-auto case (k + 1 + R > 64)
    ("float64", "rbf", 64, 40, 16, "auto", 1200),
    ("float64", "matern15", 50, 8, 16, "auto", 1200),
    ("float64", "matern05", 33, 6, 5, "rhs", 1200),
    ("float64", "matern25", 64, 12, 8, "auto", 66_000),
]


@pytest.mark.parametrize("case", BACK_CASES, ids=[f"{c[0]}-{c[1]}-k{c[2]}-d{c[3]}-R{c[4]}-{c[5]}-b{c[6]}" for c in BACK_CASES])
def test_rhs_prediction_variant_matches_oracle(case):
    """``want_ykinvy=False`` with 5..16 responses selects the back-substitution instantiation of the rhs-column
    kernel -- a different instantiation from the one every ``want_ykinvy=True`` test reaches.  Against the oracle,
    after a launch that leaves NaN bits in every resident workgroup's LDS (the multiplier rows behind k are
    never written: round-3 advisor finding)."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    dtype, kernel, k, d, R, path, b = case
    td = getattr(torch, dtype)
    rng = np.random.default_rng(1000 + BACK_CASES.index(case))
    N = 20_000
    X = rng.normal(size=(N, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, N, size=b)
    ni = rng.integers(0, N - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])
    metric = "F2" if kernel == "rbf" else "l2"
    ls = float(np.sqrt(2 * d)) if metric == "l2" else float(np.sqrt(np.sqrt(2 * d)) * 1.5)
    spec_o = orc.Spec(kernel, metric, ls, 1e-2)
    Xd, Yd, bid, nid = to_dev(X, td), to_dev(Y, td), to_dev(bi), to_dev(ni)
    other = torch.float64 if td == torch.float32 else torch.float32
    poison = torch.full((N, 64), float("nan"), device="cuda", dtype=other)
    posterior_mean_var(KernelSpec(kernel, metric, ls, 1e-2), poison, poison, bid, nid, Yd.to(other), packed=False)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec(kernel, metric, ls, 1e-2), Xd, Xd, bid, nid, Yd, info=info, path=path,
                                   packed=False)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    if path == "rhs" or k + 1 + R > 64:
        assert served.startswith((f"mgp::fused_rhs_kernel<{'float' if dtype == 'float32' else 'double'},16,true", "mgp::fused_rhs_mf_kernel<16")), served
    assert int(info.item()) == 0
    assert bool(torch.isfinite(mean).all()) and bool(torch.isfinite(var).all()), f"non-finite outputs from {served}"
    pick = np.arange(b) if b <= 1500 else rng.choice(b, size=1500, replace=False)
    if b > 1500:
        pick[:4] = [0, 1, b - 2, b - 1]
    m_ref, v_ref = orc.posterior_mean_var(spec_o, X, X, bi[pick], ni[pick], Y)
    rtol = RTOL[dtype]
    assert_close(mean.cpu().numpy()[pick], m_ref.reshape(len(pick), R), rtol, f"mean [{served}]")
    assert_rel_close(var.cpu().numpy()[pick], v_ref, rtol, f"var [{served}]")
    # the same call with y^T K^-1 y requested goes through the forward-only instantiation: same answers
    m2, v2, _ = posterior_mean_var(KernelSpec(kernel, metric, ls, 1e-2), Xd, Xd, bid, nid, Yd, want_ykinvy=True,
                                   path=path, packed=False)
    torch.cuda.synchronize()
    assert_close(mean.cpu().numpy(), m2.cpu().numpy(), 10 * rtol if dtype == "float32" else rtol, "BACK vs forward-only")
    assert_close(var.cpu().numpy(), v2.cpu().numpy(), 10 * rtol if dtype == "float32" else rtol, "BACK vs forward-only (var)")


@pytest.mark.parametrize("b", [1, 2, 3, 7, 1201])
@pytest.mark.parametrize("shape", [(64, 40, 16), (37, 24, 7), (64, 48, 16), (64, 64, 16), (33, 8, 16)], ids=lambda s: f"k{s[0]}-d{s[1]}-R{s[2]}")
def test_rhs_prediction_variants_by_shape(shape, b):
    """The fp32 prediction variants of the rhs-column kernel by shape: the kernel on the matrix cores' layout
    (mgp_fused_rhs_mf.hip) everywhere except rows of 41..48 features, which the folded variant takes -- it eliminates the
    tasks of a launch in PAIRS (an odd last task is paired with itself and written once) -- and, for a single task of
    that width, the three-wave variant.  Odd and tiny batches against the oracle."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, d, R = shape
    rng = np.random.default_rng(77 + b + k)
    N = 5_000
    X = rng.normal(size=(N, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, N, size=b)
    ni = rng.integers(0, N - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])
    ls = float(np.sqrt(2 * d))
    Xd, Yd, bid, nid = to_dev(X, torch.float32), to_dev(Y, torch.float32), to_dev(bi), to_dev(ni)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec("matern25", "l2", ls, 1e-2), Xd, Xd, bid, nid, Yd, info=info, path="rhs", packed=False)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    if d != 48 or b < 2:
        assert served == "mgp::fused_rhs_mf_kernel<16>", served
    else:
        assert served.startswith("mgp::fused_rhs_kernel<float,16,true,true") and served.endswith("fold>"), served
    assert int(info.item()) == 0
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern25", "l2", ls, 1e-2), X, X, bi, ni, Y)
    assert_close(mean.cpu().numpy(), m_ref.reshape(b, R), RTOL["float32"], f"mean [{served}]")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var [{served}]")
    # a non-positive-definite neighbourhood in the FIRST and one in the SECOND half of a pair: NaN rows, counted, neighbours untouched
    if b >= 7:
        ni2 = ni.copy()
        ni2[2, 1] = ni2[2, 0]  # duplicate neighbour with a tiny nugget: singular
        ni2[5, 3] = ni2[5, 2]
        info.zero_()
        m3, v3 = posterior_mean_var(KernelSpec("matern25", "l2", ls, 0.0), Xd, Xd, bid, to_dev(ni2), Yd, info=info, path="rhs", packed=False)
        torch.cuda.synchronize()
        m0, v0 = posterior_mean_var(KernelSpec("matern25", "l2", ls, 0.0), Xd, Xd, bid, nid, Yd, path="rhs", packed=False)
        torch.cuda.synchronize()
        ok = np.ones(b, dtype=bool)
        ok[[2, 5]] = False
        assert int(info.item()) >= 0
        np.testing.assert_array_equal(m3.cpu().numpy()[ok], m0.cpu().numpy()[ok])
        np.testing.assert_array_equal(v3.cpu().numpy()[ok], v0.cpu().numpy()[ok])


@pytest.mark.parametrize("d", [40, 48, 64, 12])
@pytest.mark.parametrize("kernel,metric", [("rbf", "F2"), ("matern25", "l2"), ("matern15", "l2")])  # (a Matern of SQUARED distances is not positive definite in general)
def test_rhs_prediction_variants_anisotropy_noise_table_and_query_table(d, kernel, metric):
    """Anisotropy, a per-training-point noise table, a separate table of queries and k short of the 64 slots through the
    fp32 prediction variants (matrix-core layout: d = 40, 64, 12; folded: d = 48), against the oracle."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, R, b, N, M = 57, 11, 515, 6_000, 900
    rng = np.random.default_rng(100 + d)
    X = rng.normal(size=(N, d))
    Q = rng.normal(size=(M, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, M, size=b)
    ni = np.stack([rng.choice(N, size=k, replace=False) for _ in range(b)])
    ls = np.sqrt(2 * d) * 10.0 ** rng.uniform(-0.2, 0.2, size=d)
    if metric == "F2":
        ls = np.sqrt(ls) * 1.5
    eps = 10.0 ** rng.uniform(-3, -1.5, size=N)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec(kernel, metric, [float(v) for v in ls], to_dev(eps, torch.float32)), to_dev(Q, torch.float32),
                                   to_dev(X, torch.float32), to_dev(bi), to_dev(ni), to_dev(Y, torch.float32), info=info, packed=False)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    assert served.endswith("fold>") if d == 48 else served == "mgp::fused_rhs_mf_kernel<16>", served
    assert int(info.item()) == 0
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec(kernel, metric, ls, eps), Q, X, bi, ni, Y)
    assert_close(mean.cpu().numpy(), m_ref.reshape(b, R), RTOL["float32"], f"mean [{served}]")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var [{served}]")


@pytest.mark.parametrize("seed", range(12))
def test_rhs_prediction_variants_random_shapes(seed):
    """Seeded random shapes through the fp32 prediction path (`path="rhs"`: nn_count 2..64, 8..64 features in whole
    16-byte groups, 5..16 responses, homoscedastic or per-point noise, shared or separate query table), against the
    oracle -- whichever variant the dispatcher picks."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    rng = np.random.default_rng(9000 + seed)
    k = int(rng.integers(2, 65))
    d = 4 * int(rng.integers(2, 17))
    R = int(rng.integers(5, 17))
    b = int(rng.integers(1, 400))
    N, M = 3_000, 500
    kernel = ["rbf", "matern15", "matern25", "maternInf"][seed % 4]
    metric = "F2" if kernel == "rbf" else "l2"
    X = rng.normal(size=(N, d))
    separate = bool(seed % 3 == 0)
    Q = rng.normal(size=(M, d)) if separate else X
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, Q.shape[0], size=b)
    ni = np.stack([rng.choice(N, size=k, replace=False) for _ in range(b)])
    if not separate:
        ni = np.where(ni == bi[:, None], (ni + 1) % N, ni)
    ls = float(np.sqrt(2 * d)) if metric == "l2" else float(np.sqrt(np.sqrt(2 * d)) * 1.5)
    eps = 10.0 ** rng.uniform(-3, -1.5, size=N) if seed % 2 else 1e-2
    noise = to_dev(eps, torch.float32) if seed % 2 else eps
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec(kernel, metric, ls, noise), to_dev(Q, torch.float32), to_dev(X, torch.float32), to_dev(bi),
                                   to_dev(ni), to_dev(Y, torch.float32), info=info, path="rhs", packed=False)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    assert served.startswith(("mgp::fused_rhs_kernel<float,16,true", "mgp::fused_rhs_mf_kernel<16")), served
    assert int(info.item()) == 0, served
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec(kernel, metric, ls, eps), Q, X, bi, ni, Y)
    assert_close(mean.cpu().numpy(), m_ref.reshape(b, R), RTOL["float32"], f"mean [{served}; k={k} d={d} R={R} b={b}]")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var [{served}; k={k} d={d} R={R} b={b}]")


def test_rhs_three_wave_variant_serves_tables_off_the_16_byte_grid():
    """A feature table whose rows do not start on 16-byte boundaries (a view one element into a larger buffer): neither
    the matrix-core kernel nor the folded variant takes it (their gathers are 16-byte transfers); the three-wave variant
    does, with its element-wise gather."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, d, R, b, N = 64, 40, 16, 333, 4_000
    rng = np.random.default_rng(5)
    X = rng.normal(size=(N, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, N, size=b)
    ni = rng.integers(0, N - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])
    ls = float(np.sqrt(2 * d))
    flat = torch.zeros(N * d + 1, device="cuda", dtype=torch.float32)
    flat[1:] = to_dev(X, torch.float32).reshape(-1)
    Xd = flat[1:].view(N, d)
    assert Xd.data_ptr() % 16 == 4
    mean, var = posterior_mean_var(KernelSpec("matern25", "l2", ls, 1e-2), Xd, Xd, to_dev(bi), to_dev(ni), to_dev(Y, torch.float32),
                                   path="rhs", packed=False)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    assert served.endswith("w3>"), served
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern25", "l2", ls, 1e-2), X, X, bi, ni, Y)
    assert_close(mean.cpu().numpy(), m_ref.reshape(b, R), RTOL["float32"], f"mean [{served}]")
    assert_rel_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var [{served}]")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_config5_fixture_through_the_prediction_variant(dtype):
    """The reference-generated config-5 fixture (RBF, k = 64, R = 16, d = 40) through the exact call bench.py's
    config-5 line makes: no y^T K^-1 y, dispatcher's choice."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import posterior_mean_var
    from tests.conftest import load_golden

    g = load_golden("rbf_iso_R16_k64_d40_c5")
    meta = g["meta"]
    td = getattr(torch, dtype)
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(_kspec(meta, g, td), X, X, to_dev(g["batch_idx"]), to_dev(g["nn_idx"]), y, info=info)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    assert served.startswith("mgp::fused_rhs_mf_kernel<16") or ("fused_rhs_kernel" in served and ",16,true" in served), served
    assert int(info.item()) == 0
    assert_close(mean.cpu().numpy(), g["mean"], RTOL[dtype], "mean")
    assert_close(var.cpu().numpy(), g["var_unscaled"], RTOL[dtype], "var")
    assert_rel_close(var.cpu().numpy(), g["var_unscaled"], RTOL[dtype], "var (relative)",
                     calibration=fp32_reference(meta["name"])[1] if dtype == "float32" else None)


PERSISTENT_CASES = [
    # dtype, kernel, k, d, R, aniso  -- run-time shapes; b is large enough that every workgroup
    # loops over many tasks (the software-pipelined gather crosses task boundaries)
    ("float32", "matern15", 29, 40, 1, False),
    ("float32", "matern25", 20, 36, 1, False),   # d % 8 == 4: zero-filled padding slot
    ("float32", "rbf", 40, 16, 3, False),        # NP = 64, several responses
    ("float32", "matern15", 12, 8, 2, True),
    ("float64", "matern15", 20, 16, 1, False),
    ("float64", "matern05", 33, 6, 1, True),     # d % 4 == 2 in fp64
    ("float32", "matern15", 30, 20, 1, False),
    ("float32", "matern15", 17, 3, 1, False),    # unaligned rows: register-staged gather
]


@pytest.mark.parametrize("case", PERSISTENT_CASES, ids=[f"{c[0]}-{c[1]}-k{c[2]}-d{c[3]}-R{c[4]}-{'aniso' if c[5] else 'iso'}" for c in PERSISTENT_CASES])
@pytest.mark.parametrize("packed", [False, True])
def test_persistent_loop_matches_oracle_on_a_sample(case, packed):
    from muygpys_amd.fused import KernelSpec, PackedTable, posterior_mean_var

    dtype, kernel, k, d, R, aniso = case
    td = getattr(torch, dtype)
    rng = np.random.default_rng(PERSISTENT_CASES.index(case))
    N, b = 50_000, 240_000
    X = rng.normal(size=(N, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, N, size=b)
    ni = rng.integers(0, N, size=(b, k))
    ni = np.where(ni == bi[:, None], (ni + 1) % N, ni)
    metric = "F2" if kernel == "rbf" else "l2"
    ls = np.sqrt(d) * rng.uniform(0.8, 1.4, size=d) if aniso else float(np.sqrt(2 * d))
    if metric == "F2":
        ls = np.sqrt(ls)
    spec_o = orc.Spec(kernel, metric, ls, 1e-2)
    if packed and not PackedTable.supported(d, R, k, td):
        pytest.skip("shape outside the prepared-table kernels")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    Xd = to_dev(X, td)
    mean, var, yk = posterior_mean_var(
        KernelSpec(kernel, metric, ls.tolist() if aniso else ls, 1e-2), Xd, Xd, to_dev(bi),
        to_dev(ni), to_dev(Y, td), want_ykinvy=True, info=info, packed=packed,
    )
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    pick = rng.choice(b, size=1500, replace=False)
    pick[:4] = [0, 1, b - 2, b - 1]
    m_ref, v_ref = orc.posterior_mean_var(spec_o, X, X, bi[pick], ni[pick], Y)
    rtol = RTOL[dtype]
    assert_close(mean.cpu().numpy()[pick], m_ref.reshape(len(pick), R), rtol, "mean")
    assert_rel_close(var.cpu().numpy()[pick], v_ref, rtol, "var")
    assert torch.isfinite(mean).all() and torch.isfinite(var).all() and torch.isfinite(yk).all()


WIDE_CASES = [
    # kernel, k, d, R, aniso  -- more rows than a wave has lanes
    ("matern15", 80, 16, 1, False),
    ("rbf", 100, 40, 2, False),
    ("matern25", 126, 8, 1, True),
    ("matern05", 64, 12, 1, False),   # 66 rows, k = 64: the rhs-column kernel
    ("matern15", 65, 12, 1, False),   # 67 rows, k > 64: the first shape of the 128-slot kernel
    ("matern15", 75, 24, 1, False),
    ("matern15", 90, 37, 3, False),   # unaligned rows, several responses
    # round 5 -- fused_wide_kernel<NG> / fused_wide64_kernel<NB>: one shape per column-group count, most of them at
    # the last row count the instantiation serves (no padding slot between the neighbours and the query)
    ("matern15", 66, 8, 1, False),    # 68 rows  -> 17 groups (fp64: 18 blocks)
    ("rbf", 67, 20, 1, True),         # 69 rows  -> 20
    ("matern25", 78, 12, 1, False),   # 80 rows  -> 20, full
    ("matern15", 84, 16, 3, False),   # 88 rows  -> 22, full
    ("matern15", 94, 8, 1, True),     # 96 rows  -> 24, full
    ("matern15", 101, 12, 2, False),  # 104 rows -> 26, full
    ("rbf", 95, 8, 16, False),        # 112 rows -> 28, full, sixteen responses
    ("matern15", 110, 8, 9, False),   # 120 rows -> 30, full
    ("matern25", 111, 8, 16, False),  # 128 rows -> 32, full, sixteen responses
    ("matern15", 65, 8, 16, False),   # 82 rows, a pair ring of 66 rows under 22 column groups
]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("case", WIDE_CASES, ids=[f"{c[0]}-k{c[1]}-d{c[2]}-R{c[3]}-{'aniso' if c[4] else 'iso'}" for c in WIDE_CASES])
def test_wide_neighbourhoods_match_oracle(case, dtype):
    """More than 64 slots: the rhs-column kernel (k <= 64), the 128-slot kernels (fp32: one lane per
    row; fp64: two lanes per row), against the oracle."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    td = getattr(torch, dtype)

    kernel, k, d, R, aniso = case
    rng = np.random.default_rng(40 + WIDE_CASES.index(case))
    N, b = 6000, 1200
    X = rng.normal(size=(N, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(N, R))
    bi = rng.integers(0, N, size=b)
    ni = np.stack([rng.choice(N, size=k, replace=False) for _ in range(b)])
    metric = "F2" if kernel == "rbf" else "l2"
    ls = np.sqrt(d) * rng.uniform(0.8, 1.4, size=d) if aniso else float(np.sqrt(2 * d))
    if metric == "F2":
        ls = np.sqrt(ls)
    spec_o = orc.Spec(kernel, metric, ls, 1e-2)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var, yk = posterior_mean_var(
        KernelSpec(kernel, metric, ls.tolist() if aniso else ls, 1e-2), to_dev(X, td),
        to_dev(X, td), to_dev(bi), to_dev(ni), to_dev(Y, td), want_ykinvy=True, info=info,
    )
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    pick = rng.choice(b, size=150, replace=False)
    m_ref, v_ref = orc.posterior_mean_var(spec_o, X, X, bi[pick], ni[pick], Y)
    Kc, Kin = orc.kernel_tensors(spec_o, orc.crosswise_tensor(X, X, bi[pick], ni[pick]), orc.pairwise_tensor(X, ni[pick]))
    Kin = orc.perturb(spec_o, Kin, ni[pick])
    yk_ref = np.einsum("bkr,bkr->br", Y[ni[pick]], np.linalg.solve(Kin, Y[ni[pick]]))
    assert_close(mean.cpu().numpy()[pick].reshape(150, R), m_ref.reshape(150, R), RTOL[dtype], "mean")
    assert_rel_close(var.cpu().numpy()[pick], v_ref, RTOL[dtype], "var")
    assert_close(yk.cpu().numpy()[pick].reshape(150, R), yk_ref, RTOL[dtype], "ykinvy")


STALE_LDS_CASES = [
    # k, R, dtype -- the 128-slot kernel (partial / full last block), the packed 64-slot exchange matrix,
    # the square exchange matrices of the 32-slot and the rhs-column kernels
    (126, 1, "float32"), (125, 1, "float32"), (122, 3, "float32"), (99, 2, "float32"),
    (65, 1, "float32"), (77, 1, "float32"), (83, 3, "float32"), (90, 16, "float32"),     # ... with padding slots under 17 / 20 / 22 / 28 groups
    (50, 1, "float32"), (50, 1, "float64"), (61, 1, "float32"), (30, 1, "float32"), (29, 2, "float64"),
    (64, 16, "float32"), (63, 4, "float64"),
    (126, 1, "float64"), (125, 1, "float64"), (99, 2, "float64"), (65, 1, "float64"),   # two lanes per row
    (48, 16, "float32"), (50, 16, "float32"), (56, 12, "float32"), (50, 16, "float64"),  # rhs columns, back-substitution variant
]


@pytest.mark.parametrize("k,R,dtype", STALE_LDS_CASES)
def test_register_kernels_ignore_stale_lds(k, R, dtype):
    """The exchange matrices of the register kernels have slots nobody writes (row padding, the upper
    triangle a lane over-reads, the pad behind the last row); they hold whatever an earlier launch left in
    LDS.  A launch on NaN features (d = 64: its feature tile covers the exchange region) poisons them; the
    results of the next launch must not change.  (The 128-slot kernel used to multiply such a slot by 0 in
    a partial last block -- k not a multiple of 4 -- and returned NaN variances, depending on what ran before.)"""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    td = getattr(torch, dtype)
    rng = np.random.default_rng(k * 7 + R)
    N, b, d = 8000, 4096, 8
    X = torch.from_numpy(rng.normal(size=(N, d))).to("cuda", td)
    Y = torch.from_numpy(rng.normal(size=(N, R))).to("cuda", td)
    bi = torch.from_numpy(rng.integers(0, N, size=b)).cuda()
    ni = torch.from_numpy(np.stack([rng.choice(N, size=k, replace=False) for _ in range(b)])).cuda()
    spec = KernelSpec("matern25", "l2", 3.0, 1e-2)
    poison = torch.full((N, 64), float("nan"), device="cuda", dtype=td)
    ref = None
    for attempt in range(3):
        posterior_mean_var(spec, poison, poison, bi, ni, Y, packed=False)  # NaN rows through every resident workgroup's LDS
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        mean, var = posterior_mean_var(spec, X, X, bi, ni, Y, info=info, packed=False)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        assert bool(torch.isfinite(mean).all()) and bool(torch.isfinite(var).all()), f"NaN leaked (attempt {attempt})"
        if ref is None:
            ref = (mean.clone(), var.clone())
        assert torch.equal(mean, ref[0]) and torch.equal(var, ref[1])


def _random_shapes():
    rng = np.random.default_rng(20261003)
    shapes = []
    for _ in range(36):
        k = int(rng.choice([3, 7, 12, 20, 29, 30, 31, 33, 47, 50, 61, 62, 63, 64, 65, 70, 90, 100, 111, 126]))
        R = int(rng.choice([1, 1, 2, 3, 5, 16]))
        if k + 1 + R > 128:
            R = 1
        d = int(rng.choice([1, 3, 4, 8, 10, 17, 24, 40, 64]))
        dtype = str(rng.choice(["float32", "float64"]))
        kernel = str(rng.choice(["rbf", "matern05", "matern15", "matern25", "maternInf"]))
        aniso = bool(rng.integers(0, 2))
        shapes.append((k, R, d, dtype, kernel, aniso))
    return shapes


@pytest.mark.parametrize("shape", _random_shapes(), ids=lambda s: f"k{s[0]}-R{s[1]}-d{s[2]}-{s[3]}-{s[4]}-{'aniso' if s[5] else 'iso'}")
def test_dispatcher_agrees_with_the_lds_workgroup_kernel(shape):
    """Random shapes across every dispatch boundary (32 / 64 / 128 slots, rhs columns, fp64 two-lane rows,
    aligned and unaligned rows): whichever register kernel the dispatcher picks agrees with the
    independent LDS workgroup kernel on the same inputs."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, R, d, dtype, kernel, aniso = shape
    td = getattr(torch, dtype)
    rng = np.random.default_rng(k * 131 + R * 17 + d)
    N, b = 3000, 777
    X = torch.from_numpy(rng.normal(size=(N, d))).to("cuda", td)
    Y = torch.from_numpy(np.sin(rng.normal(size=(N, R)))).to("cuda", td)
    bi = torch.from_numpy(rng.integers(0, N, size=b)).cuda()
    ni = torch.from_numpy(np.stack([rng.choice(N, size=k, replace=False) for _ in range(b)])).cuda()
    metric = "F2" if kernel == "rbf" else "l2"
    base = np.sqrt(2.0 * d) if metric == "l2" else np.sqrt(np.sqrt(2.0 * d))
    ls = (base * rng.uniform(0.8, 1.3, size=d)).tolist() if aniso else float(base)
    spec = KernelSpec(kernel, metric, ls, 1e-2)
    Yt = Y[:, 0] if R == 1 else Y
    m1, v1, y1 = posterior_mean_var(spec, X, X, bi, ni, Yt, want_ykinvy=True, path="auto", packed=False)
    m2, v2, y2 = posterior_mean_var(spec, X, X, bi, ni, Yt, want_ykinvy=True, path="generic")
    torch.cuda.synchronize()
    rtol = 10 * RTOL[dtype] if dtype == "float32" else RTOL[dtype]  # two fp32 kernels against each other
    assert_close(m1.cpu().numpy().reshape(b, R), m2.cpu().numpy().reshape(b, R), rtol, f"mean [{_lib.served_by(d, k, R, td, False, 'auto')}]")
    assert_close(v1.cpu().numpy(), v2.cpu().numpy(), rtol, "var")
    assert_close(y1.cpu().numpy().reshape(b, R), y2.cpu().numpy().reshape(b, R), rtol, "ykinvy")


@pytest.mark.parametrize("k,d,R,b", [(64, 40, 16, 5003), (56, 24, 12, 1001), (60, 8, 16, 257), (64, 64, 5, 130)])
def test_prepared_tables_with_up_to_sixteen_responses_give_identical_bits(k, d, R, b):
    """Round 5 (BASELINE config 5): rows [features | 16 responses | pad] at a 64-byte multiple stride (256 B at d = 40)
    read by the fp32 prediction kernel on the matrix cores' layout -- a neighbour's responses come with the second
    128-byte line of its feature row instead of from a third line of a separate response tensor
    (reference gather: gp/muygps.py:474).  Same kernel, same arithmetic: the same bits as from the plain tables; and
    the oracle's values."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, PackedTable, posterior_mean_var

    rng = np.random.default_rng(500 + k + R)
    n = 3000
    X, Q = rng.normal(size=(n, d)), rng.normal(size=(700, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.05 * rng.normal(size=(n, R))
    bi = rng.integers(0, 700, size=b)
    ni = np.stack([rng.choice(n, size=k, replace=False) for _ in range(b)])
    assert PackedTable.supported(d, R, k, torch.float32)
    Xd, Qd, Yd = to_dev(X, torch.float32), to_dev(Q, torch.float32), to_dev(Y, torch.float32)
    spec = KernelSpec("rbf", "F2", 4.0 if d >= 24 else 2.0, 1e-2)
    out = {}
    for packed in (True, False):
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        mean, var = posterior_mean_var(spec, Qd, Xd, to_dev(bi), to_dev(ni), Yd, info=info, packed=packed)
        torch.cuda.synchronize()
        served = _lib.last_kernel()
        assert served.startswith("mgp::fused_rhs_mf_kernel<16>") and served.endswith("[prepared tables]") == packed, served
        assert int(info.item()) == 0
        out[packed] = (mean.clone(), var.clone())
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    pick = np.arange(min(b, 400))
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("rbf", "F2", spec.length_scale, 1e-2), Q, X, bi[pick], ni[pick], Y)
    assert_close(out[True][0].cpu().numpy()[pick], m_ref.reshape(len(pick), R), RTOL["float32"], "mean")
    assert_rel_close(out[True][1].cpu().numpy()[pick], v_ref, RTOL["float32"], "var")
