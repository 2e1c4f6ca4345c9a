"""One LOOCV objective evaluation = ONE launch (round 5): a workgroup of the fused wave kernel that has run out of tasks
reduces its own outputs -- its leaf of a fixed reduction tree -- and the last arrivers walk up (csrc/mgp_loocv_tree.h).
Pinned here:

* the sums equal -- bit for bit -- the same tree walked by three small launches over the finished outputs
  (mgp_loocv_tree_* on the same leaves), for every kernel family that serves a LOOCV call, whatever workgroup finished
  when;
* they equal the oracle's losses / sigma^2 (reference: _src/optimize/loss/numpy.py:22-72, scale/numpy.py:9-34);
* the scratch's counters are left zero: the same buffer serves call after call;
* the outputs equal the prediction launch's, bit for bit (the walk changes nothing in the task loop).
"""

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a ROCm device")]


def _problem(seed, n, b, k, d, dtype):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d))
    y = np.sin(X @ rng.normal(size=d) / np.sqrt(d)) + 0.1 * rng.normal(size=n)
    bi = rng.choice(n, size=b, replace=b > n)
    ni = rng.integers(0, n - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])  # never the batch row itself
    td = getattr(torch, dtype)
    return X, y, bi, ni, to_dev(X, td), to_dev(y, td), to_dev(bi), to_dev(ni)


# (dtype, k, d, packed, b): the built-in static shapes (folded pairs / dealt triangle), run-time compiled ones,
# run-time-shape kernels with 4 / 2 / 1 neighbourhoods per wave, and shapes the wave kernels do not serve (the tree is
# then walked behind the rhs-column kernel).  Batch sizes on and off every block boundary of the tree.
CASES = [
    ("float32", 30, 40, True, 1), ("float32", 30, 40, True, 63), ("float32", 30, 40, True, 64), ("float32", 30, 40, True, 65),
    ("float32", 30, 40, True, 4097), ("float32", 30, 40, True, 70_001), ("float32", 30, 40, "auto", 262_147),
    ("float32", 30, 40, False, 9_999),
    ("float64", 50, 8, True, 4_100), ("float64", 50, 8, True, 66_003), ("float64", 30, 40, True, 8_200),
    ("float32", 10, 8, True, 5_003), ("float32", 20, 16, False, 66_000), ("float32", 45, 24, True, 4_099),
    ("float64", 12, 6, False, 3_001), ("float32", 62, 16, True, 2_050),
    ("float32", 64, 40, False, 1_030), ("float64", 70, 8, False, 517),
]


@pytest.mark.parametrize("dtype,k,d,packed,b", CASES, ids=lambda v: str(v))
def test_one_launch_sums_equal_the_tree_walked_by_kernels_and_the_oracle(dtype, k, d, packed, b):
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, loocv_partials, loocv_tree_sums, posterior_mean_var

    n = 5_000
    X, y, bi, ni, Xd, yd, bid, nid = _problem(11 + k + b % 97, n, b, k, d, dtype)
    spec = KernelSpec("matern15", "l2", 3.0 if d >= 16 else 1.5, 1e-2)
    runs = []
    for rep in range(3):  # (the same scratch: every call must leave its counters zero)
        p, mean, var, yk = loocv_partials(spec, Xd, yd, bid, nid, huber_delta=1.5, packed=packed, return_ykinvy=True)
        served = _lib.last_kernel()
        torch.cuda.synchronize()
        runs.append((p.cpu().numpy().copy(), mean.cpu().numpy().copy(), var.cpu().numpy().copy()))
    for p, m, v in runs[1:]:
        assert np.array_equal(p.view(np.int64), runs[0][0].view(np.int64)), f"sums differ between calls [{served}]: {p} vs {runs[0][0]}"
        assert np.array_equal(m, runs[0][1]) and np.array_equal(v, runs[0][2])
    p0 = runs[0][0]
    # the tree walked by the three small launches over the call's own outputs, on the same leaves (the fused launch's
    # workgroups; (0, 0) when another kernel family served the call and the canonical leaves were walked), returns the
    # same sums, bit for bit
    leaves = _lib.last_loocv_geometry()
    assert (leaves[0] > 0) == served.startswith("mgp::fused_wave_kernel"), (leaves, served)
    p_tree = loocv_tree_sums(mean, var, yk, yd, bid, 1.5, leaves=leaves).cpu().numpy()
    assert np.array_equal(p_tree.view(np.int64), p0.view(np.int64)), f"[{served}] in-kernel {p0} vs kernels {p_tree}"
    # (other leaves: the same sums to fp64 rounding)
    np.testing.assert_allclose(loocv_tree_sums(mean, var, yk, yd, bid, 1.5).cpu().numpy(), p0, rtol=1e-12, atol=1e-9)
    # the prediction launch: the same instantiation, the same bits per neighbourhood
    m2, v2, yk2 = posterior_mean_var(spec, Xd, Xd, bid, nid, yd, want_ykinvy=True, packed=packed)
    torch.cuda.synchronize()
    served_pred = _lib.last_kernel()
    if served == served_pred:
        assert np.array_equal(m2.cpu().numpy(), runs[0][1]) and np.array_equal(v2.cpu().numpy(), runs[0][2]), (served, served_pred)
        assert torch.equal(yk2, yk)
    else:
        assert_close(m2.cpu().numpy(), runs[0][1], 10 * np.finfo(getattr(np, dtype)).eps ** 0.75, f"mean [{served} vs {served_pred}]")
    yk2 = yk
    assert p0[3] == b
    # oracle (fp64 numpy restatement of the reference)
    pick = np.arange(b) if b <= 3000 else np.random.default_rng(5).choice(b, size=3000, replace=False)
    ospec = orc.Spec("matern15", "l2", spec.length_scale, 1e-2)
    m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi[pick], ni[pick], y)
    assert_close(runs[0][1][pick], m_ref, RTOL[dtype], f"mean [{served}]")
    # the sums against the per-neighbourhood outputs in numpy fp64
    mm, vv = runs[0][1].astype(np.float64), runs[0][2].astype(np.float64)
    r = mm - y.astype(getattr(np, dtype)).astype(np.float64)[bi]
    ref = np.array([np.sum(r * r / vv), np.sum(np.log(vv)), np.sum(r * r), b,
                    np.sum(1.5**2 * (np.sqrt(1 + (r / 1.5) ** 2) - 1)), np.sum(yk2.cpu().numpy().astype(np.float64))])
    np.testing.assert_allclose(p0, ref, rtol=1e-11, atol=1e-9)


def test_scratch_is_reusable_across_batch_sizes_and_interleaved_shapes():
    """The Python layer keeps one zeroed scratch per (stream, batch size); alternating batch sizes and kernels must not
    see each other's counters."""
    from muygpys_amd.fused import KernelSpec, loocv_partials

    spec = KernelSpec("matern15", "l2", 3.0, 1e-2)
    ref = {}
    for rep in range(3):
        for b in (200, 4_097, 200, 33_000):
            _, _, _, _, Xd, yd, bid, nid = _problem(3, 4_000, b, 30, 40, "float32")
            p, _, _ = loocv_partials(spec, Xd, yd, bid, nid, packed=True)
            torch.cuda.synchronize()
            got = p.cpu().numpy()
            assert got[3] == b
            if b in ref:
                assert np.array_equal(got.view(np.int64), ref[b].view(np.int64)), (rep, b, got, ref[b])
            ref[b] = got


def test_one_launch_under_uneven_load_with_warm_caches():
    """Hand-offs must hold when the reducing CU has the handed-off lines in its L1 (it READ them before they were
    rewritten) and while other work runs: fill mean / var / ykinvy with garbage through plain loads and stores of a
    torch kernel, start a long unrelated kernel on a second stream, then evaluate; many times."""
    from muygpys_amd.fused import KernelSpec, loocv_partials

    spec = KernelSpec("matern15", "l2", 3.0, 1e-2)
    _, _, _, _, Xd, yd, bid, nid = _problem(8, 20_000, 150_001, 30, 40, "float32")
    p_ref, m_ref, v_ref = loocv_partials(spec, Xd, yd, bid, nid, packed=True)
    torch.cuda.synchronize()
    p_ref = p_ref.cpu().numpy()
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device="cuda")
    for rep in range(25):
        with torch.cuda.stream(side):
            for _ in range(3):
                junk.mul_(1.0000001)
        p, m, v = loocv_partials(spec, Xd, yd, bid, nid, packed=True)
        torch.cuda.synchronize()
        assert np.array_equal(p.cpu().numpy().view(np.int64), p_ref.view(np.int64)), (rep, p.cpu().numpy(), p_ref)
        assert torch.equal(m, m_ref) and torch.equal(v, v_ref)


@pytest.mark.parametrize("dtype,k,d,aniso", [("float32", 30, 40, False), ("float64", 50, 8, True), ("float32", 20, 16, True),
                                             ("float64", 12, 6, False)])
def test_prepared_evaluation_equals_the_plain_call_bit_for_bit(dtype, k, d, aniso):
    """fused.LoocvPlan (tables, buffers and argument list set up once; the six sums written by the kernel to pinned host
    memory and polled) against fused.loocv_partials, over several hyper-parameter points and both result routes."""
    from muygpys_amd.fused import KernelSpec, LoocvPlan, loocv_partials

    b = 20_003
    X, y, bi, ni, Xd, yd, bid, nid = _problem(77 + k, 6_000, b, k, d, dtype)
    plans = [LoocvPlan("matern15", "l2", Xd, yd, bid, nid, anisotropic=aniso, host_result=h) for h in (True, False)]
    rng = np.random.default_rng(2)
    for trial in range(4):
        ls = (rng.uniform(1.0, 3.0, size=d) if aniso else float(rng.uniform(1.0, 3.0)))
        eps = float(rng.uniform(1e-3, 1e-1))
        spec = KernelSpec("matern15", "l2", ls.tolist() if aniso else ls, eps)
        p_ref, m_ref, v_ref = loocv_partials(spec, Xd, yd, bid, nid, packed=True)
        torch.cuda.synchronize()
        for plan in plans:
            got = plan.evaluate(ls, eps)
            torch.cuda.synchronize()
            assert isinstance(got, np.ndarray) and got.shape == (6,)
            assert np.array_equal(got.view(np.int64), p_ref.cpu().numpy().view(np.int64)), (trial, plan.host_result, got, p_ref)
            assert torch.equal(plan.mean, m_ref) and torch.equal(plan.var, v_ref)
            assert int(plan.info.item()) == 0


def test_sharded_loocv_uses_the_prepared_evaluation_and_matches_the_oracle_losses():
    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec

    b, k, d = 9_001, 30, 40
    X, y, bi, ni, Xd, yd, bid, nid = _problem(5, 5_000, b, k, d, "float64")
    ospec = orc.Spec("matern15", "l2", 3.0, 1e-2)
    m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi, ni, y)
    sig = orc.sigma_sq(ospec, X, ni, y) if hasattr(orc, "sigma_sq") else None
    D.clear_plans()
    for rep in range(3):
        out = D.sharded_loocv(KernelSpec("matern15", "l2", 3.0, 1e-2), Xd, yd, bid, nid, loss="lool", presharded=True)
        assert len(D._PLANS) == 1  # one prepared evaluation serves every call on these tensors
        r = m_ref - y[bi]
        s2 = out["sigma_sq"]
        if sig is not None:
            np.testing.assert_allclose(s2, np.ravel(sig)[0], rtol=1e-9)
        lool = np.sum(r * r / (s2 * v_ref) + np.log(s2 * v_ref))
        np.testing.assert_allclose(out["lool"], lool, rtol=1e-8)
        np.testing.assert_allclose(out["mse"], np.mean(r * r), rtol=1e-8)
    # another model structure on the same tensors gets its own plan
    D.sharded_loocv(KernelSpec("rbf", "F2", 3.0, 1e-2), Xd, yd, bid, nid, loss="mse", presharded=True)
    assert len(D._PLANS) == 2
    D.clear_plans()


@pytest.fixture
def tree_mode():
    """Set the hand-off form for one test and put the process's own back."""
    from muygpys_amd import _lib

    before = _lib.loocv_tree_mode()
    yield _lib.set_loocv_tree_mode
    _lib.set_loocv_tree_mode(before)


@pytest.mark.parametrize("dtype,k,d,packed,b", [("float32", 30, 40, True, 70_001), ("float64", 50, 8, True, 4_100),
                                                ("float32", 10, 8, True, 5_003), ("float32", 64, 40, False, 1_030)],
                         ids=lambda v: str(v))
def test_the_three_hand_off_forms_give_the_same_bits(dtype, k, d, packed, b, tree_mode):
    """MUYGPYS_HIP_LOOCV_TREE = tickets | fenced | three_launch (csrc/mgp_loocv_tree.h): the same leaves, the same order,
    the same sums bit for bit -- and the same outputs; ``three_launch`` reports the fused launch's leaves although the
    kernel did not walk them."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, loocv_partials

    X, y, bi, ni, Xd, yd, bid, nid = _problem(5 + k, 5_000, b, k, d, dtype)
    spec = KernelSpec("matern15", "l2", 3.0 if d >= 16 else 1.5, 1e-2)
    got = {}
    for mode in _lib.TREE_MODES:
        tree_mode(mode)
        assert _lib.loocv_tree_mode() == mode
        p, mean, var = loocv_partials(spec, Xd, yd, bid, nid, packed=packed)
        got[mode] = (p.cpu().numpy().tobytes(), mean.cpu().numpy().tobytes(), var.cpu().numpy().tobytes(),
                     _lib.last_loocv_geometry(), _lib.last_kernel())
    assert got["tickets"][:4] == got["fenced"][:4] == got["three_launch"][:4], {m: (v[3], v[4]) for m, v in got.items()}


@pytest.mark.parametrize("mode", ["tickets", "fenced"])
def test_ten_thousand_back_to_back_evaluations_equal_the_three_launch_walk(mode, tree_mode):
    """The in-kernel walk under load: 10 000 evaluations back to back on one stream (no synchronisation in between, one
    scratch buffer), at three grid sizes -- one level-2 block, a ragged grid, the full persistent grid --, every one of
    them compared with the sums of the three-launch walk of the same leaves, bit for bit.  A hand-off that a driver
    or compiler update reorders shows up here as a differing sum (or a sum that never arrives), not as a silently
    wrong loss (reference: the host reductions of _src/optimize/loss/mpi.py:57 this replaces)."""
    from muygpys_amd.fused import KernelSpec, LoocvPlan

    n, k, d = 6_000, 30, 40
    for b, evaluations in ((64, 4_000), (4_097, 3_000), (40_000, 3_000)):
        X, y, bi, ni, Xd, yd, bid, nid = _problem(3 + b, n, b, k, d, "float32")
        tree_mode("three_launch")
        ref = LoocvPlan("matern15", "l2", Xd, yd, bid, nid, host_result=False)
        want = {}
        for nz in (1e-2, 2e-2, 5e-2):   # (the noise travels by value with each launch; the length scale is one host word)
            ref.launch(3.0, nz)
            want[nz] = ref.wait().tobytes()
        tree_mode(mode)
        plan = LoocvPlan("matern15", "l2", Xd, yd, bid, nid, host_result=False)
        # device-side results, collected without a host round trip: evaluation e's six sums are copied behind it
        out = torch.zeros((evaluations, 6), device=Xd.device, dtype=torch.float64)
        noises = [(1e-2, 2e-2, 5e-2)[e % 3] for e in range(evaluations)]
        for e, nz in enumerate(noises):
            plan._in_flight = None      # (the stream orders them: same plan, same buffers, no host wait)
            plan.launch(3.0, nz)
            out[e].copy_(plan.partials, non_blocking=True)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        bad = [e for e, nz in enumerate(noises) if got[e].tobytes() != want[nz]]
        assert not bad, f"{mode}, b = {b}: {len(bad)} of {evaluations} evaluations differ from the three-launch walk (first: {bad[:5]})"
        from muygpys_amd import _lib

        zero = int(_lib.load().mgp_loocv_scratch_zero_bytes())
        assert int(plan.scratch[:zero].to(torch.int64).sum()) == 0  # (control block and counters left zero)


def test_startup_selfcheck_keeps_the_in_kernel_walk_and_falls_back_when_it_disagrees(monkeypatch, tree_mode):
    from muygpys_amd import _lib, fused

    tree_mode("tickets")
    monkeypatch.delenv("MUYGPYS_HIP_LOOCV_TREE", raising=False)
    assert _lib.loocv_tree_selfcheck(force=True) is True and _lib.loocv_tree_mode() == "tickets"
    # a walk that returns other bits than the three-launch one (simulated: the in-kernel result perturbed in its last
    # place) must switch the process over -- and the evaluation behind it is then served by the three launches
    real = fused.loocv_partials

    def flaky(*a, **kw):
        res = real(*a, **kw)
        if _lib.loocv_tree_mode() != "three_launch":
            res[0][0] = torch.nextafter(res[0][0], res[0][0] + 1)
        return res

    monkeypatch.setattr(fused, "loocv_partials", flaky)
    with pytest.warns(RuntimeWarning, match="three_launch"):
        assert _lib.loocv_tree_selfcheck(force=True) is False
    assert _lib.loocv_tree_mode() == "three_launch"
    monkeypatch.setattr(fused, "loocv_partials", real)
    _lib._TREE_CHECKED.update(done=True, fell_back=False)
