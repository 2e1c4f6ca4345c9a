"""GPU, BASELINE configs 4 and 5 at their full sizes: size-independent properties of the posterior (no oracle run is
feasible there) -- linearity in the responses, independence of the variance from them, shard concatenation bit for
bit, neighbour-permutation invariance, agreement with the independent LDS workgroup kernel on a sample, and for the
LOOCV objective the sum of shard partials = the whole batch's.  (tests/test_gpu_properties.py does the same for
configs 2 / 3.)"""

import numpy as np
import pytest

from tests.util import assert_close

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(spec, X, bi, ni, y, **kw):
    from muygpys_amd.fused import posterior_mean_var

    out = posterior_mean_var(spec, X, X, bi, ni, y, **kw)
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def c4():
    """10 M points, d = 8, k = 50, fp64, anisotropic Matern-3/2, nugget 1e-5, 2 M neighbourhoods (bench.py's config 4)."""
    from muygpys_amd.fused import KernelSpec

    g = torch.Generator(device="cuda").manual_seed(44)
    N, d, k, b = 10_000_000, 8, 50, 2_000_000
    X = torch.randn((N, d), device="cuda", dtype=torch.float64, generator=g)
    w = torch.randn((d,), device="cuda", dtype=torch.float64, generator=g) / d**0.5
    y1 = torch.sin(X @ w) + 0.1 * torch.randn((N,), device="cuda", dtype=torch.float64, generator=g)
    y2 = torch.cos(X @ w.flip(0))
    bi = torch.randperm(N, device="cuda", generator=g)[:b].contiguous()
    ni = torch.randint(0, N - 1, (b, k), device="cuda", generator=g)
    ni = ni + (ni >= bi[:, None])
    ls = (np.exp(np.random.default_rng(2).uniform(np.log(0.5), np.log(2.0), size=d)) * float(np.sqrt(d / 40.0) * 5.0)).tolist()
    yield dict(X=X, y1=y1, y2=y2, bi=bi, ni=ni, b=b, k=k, spec=KernelSpec("matern15", "l2", ls, 1e-5))
    torch.cuda.empty_cache()


def test_config4_properties_at_full_size(c4):
    from muygpys_amd import _lib
    from muygpys_amd import distributed as D

    d = c4
    m1, v1, yk1 = _run(d["spec"], d["X"], d["bi"], d["ni"], d["y1"], want_ykinvy=True)
    assert "fused_wave_kernel<double,64,50,1,8" in _lib.last_kernel(), _lib.last_kernel()
    m2, v2 = _run(d["spec"], d["X"], d["bi"], d["ni"], d["y2"])
    m3, v3 = _run(d["spec"], d["X"], d["bi"], d["ni"], 2.0 * d["y1"] - 0.5 * d["y2"])
    assert torch.equal(v1, v2) and torch.equal(v1, v3), "the variance must not depend on the responses"
    assert_close(m3.cpu().numpy(), (2.0 * m1 - 0.5 * m2).cpu().numpy(), 1e-5, "linearity")
    assert bool(torch.isfinite(m1).all() and (v1 > 0).all() and (v1 <= 1.0 + 1e-9).all() and (yk1 > 0).all())
    # shards (reference chunk rule, 3 ranks: odd boundaries) concatenate bit for bit
    parts = []
    for r in range(3):
        lo, hi = D.shard_bounds(d["b"], r, 3)
        parts.append(_run(d["spec"], d["X"], d["bi"][lo:hi].contiguous(), d["ni"][lo:hi].contiguous(), d["y1"]))
    assert torch.equal(torch.cat([p[0] for p in parts]), m1) and torch.equal(torch.cat([p[1] for p in parts]), v1)
    # a sample against the independent LDS workgroup kernel, and under a permutation of the neighbours
    sub = torch.arange(0, d["b"], 41, device="cuda")[:40_000]
    mg, vg, ykg = _run(d["spec"], d["X"], d["bi"][sub], d["ni"][sub], d["y1"], want_ykinvy=True, path="generic")
    assert_close(m1[sub].cpu().numpy(), mg.cpu().numpy(), 1e-5, "mean vs workgroup kernel")
    assert_close(v1[sub].cpu().numpy(), vg.cpu().numpy(), 1e-5, "var vs workgroup kernel")
    assert_close(yk1[sub].cpu().numpy(), ykg.cpu().numpy(), 1e-5, "ykinvy vs workgroup kernel")
    perm = torch.argsort(torch.rand((sub.numel(), d["k"]), device="cuda"), dim=1)
    mp, vp = _run(d["spec"], d["X"], d["bi"][sub], torch.gather(d["ni"][sub], 1, perm), d["y1"])
    assert_close(mp.cpu().numpy(), m1[sub].cpu().numpy(), 1e-5, "mean under permutation")
    assert_close(vp.cpu().numpy(), v1[sub].cpu().numpy(), 1e-5, "var under permutation")


def test_config4_objective_partials_add_up(c4):
    """The six fp64 partial sums of the LOOCV objective over three shards add up to the whole batch's (what the one
    all-reduce of a sharded evaluation relies on), and sigma^2 / lool are finite."""
    from muygpys_amd import distributed as D

    d = c4
    whole = D.hip_local_partials(d["spec"], d["X"], d["y1"], d["bi"], d["ni"])[0]
    acc = torch.zeros_like(whole)
    for r in range(3):
        lo, hi = D.shard_bounds(d["b"], r, 3)
        acc += D.hip_local_partials(d["spec"], d["X"], d["y1"], d["bi"][lo:hi].contiguous(), d["ni"][lo:hi].contiguous())[0]
    torch.cuda.synchronize()
    np.testing.assert_allclose(acc.cpu().numpy(), whole.cpu().numpy(), rtol=1e-12)
    out = D.finish_objective(whole.tolist(), d["k"], "lool")
    assert np.isfinite(out["lool"]) and out["sigma_sq"] > 0 and out["count"] == d["b"]


def test_config5_properties_at_full_size():
    """2 M points, d = 40, k = 64, 16 responses, RBF, fp32, 500 k neighbourhoods (bench.py's config 5): the prediction
    variant of the rhs-column kernel (no y^T K^-1 y) and the forward-only one agree, linearity, shard concatenation, the
    workgroup kernel on a sample."""
    from muygpys_amd import _lib
    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec

    g = torch.Generator(device="cuda").manual_seed(55)
    N, dd, k, R, b = 2_000_000, 40, 64, 16, 500_000
    X = torch.randn((N, dd), device="cuda", generator=g)
    W = torch.randn((dd, R), device="cuda", generator=g) / dd**0.5
    Y1 = torch.sin(X @ W) + 0.1 * torch.randn((N, R), device="cuda", generator=g)
    Y2 = torch.cos(X @ W.flip(0))
    bi = torch.randperm(N, device="cuda", generator=g)[:b].contiguous()
    ni = torch.randint(0, N - 1, (b, k), device="cuda", generator=g)
    ni = ni + (ni >= bi[:, None])
    spec = KernelSpec("rbf", "F2", 5.0, 1e-3)
    m1, v1 = _run(spec, X, bi, ni, Y1)
    assert _lib.last_kernel().startswith(("mgp::fused_rhs_mf_kernel<16", "mgp::fused_rhs_kernel<float,16,true")), _lib.last_kernel()
    m2, v2 = _run(spec, X, bi, ni, Y2)
    m3, v3 = _run(spec, X, bi, ni, 2.0 * Y1 - 0.5 * Y2)
    assert torch.equal(v1, v2) and torch.equal(v1, v3)
    assert_close(m3.cpu().numpy(), (2.0 * m1 - 0.5 * m2).cpu().numpy(), 1e-3, "linearity")
    assert bool(torch.isfinite(m1).all() and (v1 > 0).all() and (v1 <= 1.0 + 1e-6).all())
    mf, vf, ykf = _run(spec, X, bi, ni, Y1, want_ykinvy=True)  # forward-only instantiation
    assert_close(m1.cpu().numpy(), mf.cpu().numpy(), 1e-3, "prediction variant vs forward-only (mean)")
    assert_close(v1.cpu().numpy(), vf.cpu().numpy(), 1e-3, "prediction variant vs forward-only (var)")
    parts = []
    for r in range(3):
        lo, hi = D.shard_bounds(b, r, 3)
        parts.append(_run(spec, X, bi[lo:hi].contiguous(), ni[lo:hi].contiguous(), Y1))
    assert torch.equal(torch.cat([p[0] for p in parts]), m1) and torch.equal(torch.cat([p[1] for p in parts]), v1)
    sub = torch.arange(0, b, 23, device="cuda")[:20_000]
    mg, vg = _run(spec, X, bi[sub], ni[sub], Y1, path="generic")
    assert_close(m1[sub].cpu().numpy(), mg.cpu().numpy(), 1e-3, "mean vs workgroup kernel")
    assert_close(v1[sub].cpu().numpy(), vg.cpu().numpy(), 1e-3, "var vs workgroup kernel")
