"""GPU, BASELINE config-2 size (1 M points, d=40, k=30): size-independent properties of the
posterior that need no oracle run -- linearity in the responses, independence of the variance
from them, invariance under neighbour permutation, shard concatenation, agreement of the two
independent kernel implementations and of fp32 with fp64."""

import numpy as np
import pytest

from tests.util import assert_close

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N, D, K, B = 1_000_000, 40, 30, 1_000_000


@pytest.fixture(scope="module")
def data():
    from muygpys_amd.fused import KernelSpec

    g = torch.Generator(device="cuda").manual_seed(1234)
    X = torch.randn((N, D), device="cuda", dtype=torch.float32, generator=g)
    w = torch.randn((D,), device="cuda", dtype=torch.float32, generator=g) / D**0.5
    y1 = torch.sin(X @ w) + 0.1 * torch.randn((N,), device="cuda", generator=g)
    y2 = torch.cos(X @ w.flip(0)) + 0.1 * torch.randn((N,), device="cuda", generator=g)
    bi = torch.arange(B, device="cuda")
    ni = torch.randint(0, N - 1, (B, K), device="cuda", generator=g)
    ni = ni + (ni >= bi[:, None])
    return dict(X=X, y1=y1, y2=y2, bi=bi, ni=ni, spec=KernelSpec("matern15", "l2", 5.0, 1e-3))


def run(d, y, X=None, bi=None, ni=None, spec=None, **kw):
    from muygpys_amd.fused import posterior_mean_var

    out = posterior_mean_var(spec or d["spec"], d["X"] if X is None else X, d["X"] if X is None else X,
                             d["bi"] if bi is None else bi, d["ni"] if ni is None else ni, y, **kw)
    torch.cuda.synchronize()
    return out


def test_linearity_in_targets_and_variance_independence(data):
    m1, v1 = run(data, data["y1"])
    m2, v2 = run(data, data["y2"])
    m3, v3 = run(data, 2.0 * data["y1"] - 0.5 * data["y2"])
    assert torch.equal(v1, v2) and torch.equal(v1, v3), "the variance must not depend on the responses"
    assert_close(m3.cpu().numpy(), (2.0 * m1 - 0.5 * m2).cpu().numpy(), 1e-3, "linearity")
    assert bool(((v1 > 0) & (v1 <= 1.0 + 1e-6)).all()), "0 < var <= Kout"


def test_neighbour_permutation_invariance(data):
    m, v = run(data, data["y1"])
    perm = torch.argsort(torch.rand((B, K), device="cuda"), dim=1)
    ni_p = torch.gather(data["ni"], 1, perm)
    mp, vp = run(data, data["y1"], ni=ni_p)
    assert_close(mp.cpu().numpy(), m.cpu().numpy(), 1e-3, "mean under permutation")
    assert_close(vp.cpu().numpy(), v.cpu().numpy(), 1e-3, "var under permutation")


def test_shards_concatenate_exactly(data):
    from muygpys_amd import distributed as D_

    m, v = run(data, data["y1"])
    parts = []
    for r in range(3):
        lo, hi = D_.shard_bounds(B, r, 3)
        parts.append(run(data, data["y1"], bi=data["bi"][lo:hi].contiguous(), ni=data["ni"][lo:hi].contiguous()))
    assert torch.equal(torch.cat([p[0] for p in parts]), m)
    assert torch.equal(torch.cat([p[1] for p in parts]), v)


def test_wave_and_generic_kernels_agree(data):
    sub = slice(0, 200_000)
    bi, ni = data["bi"][sub].contiguous(), data["ni"][sub].contiguous()
    m, v, yk = run(data, data["y1"], bi=bi, ni=ni, want_ykinvy=True)
    mg, vg, ykg = run(data, data["y1"], bi=bi, ni=ni, want_ykinvy=True, path="generic")
    assert_close(m.cpu().numpy(), mg.cpu().numpy(), 1e-3, "mean")
    assert_close(v.cpu().numpy(), vg.cpu().numpy(), 1e-3, "var")
    assert_close(yk.cpu().numpy(), ykg.cpu().numpy(), 1e-3, "ykinvy")


def test_fp32_matches_fp64_at_full_size(data):
    m32, v32, yk32 = run(data, data["y1"], want_ykinvy=True)
    X64, y64 = data["X"].double(), data["y1"].double()
    from muygpys_amd.fused import posterior_mean_var

    m64, v64, yk64 = posterior_mean_var(data["spec"], X64, X64, data["bi"], data["ni"], y64, want_ykinvy=True)
    torch.cuda.synchronize()
    assert_close(m32.cpu().numpy(), m64.cpu().numpy(), 1e-3, "mean fp32 vs fp64")
    assert_close(v32.cpu().numpy(), v64.cpu().numpy(), 1e-3, "var fp32 vs fp64")
    s32 = float(yk32.double().sum() / (B * K))
    s64 = float(yk64.sum() / (B * K))
    assert abs(s32 - s64) <= 1e-3 * abs(s64), "sigma_sq fp32 vs fp64"


def test_query_inside_its_own_neighbourhood_interpolates(data):
    """If the query is one of its neighbours, var = eps - eps^2 [(K+eps I)^-1]_jj in (0, eps]."""
    b = 100_000
    ni = data["ni"][:b].clone()
    bi = ni[:, 7].contiguous()  # the query IS neighbour 7
    m, v = run(data, data["y1"], bi=bi, ni=ni.contiguous())
    eps = 1e-3
    assert bool((v > -1e-5).all()) and bool((v <= eps + 1e-5).all())
    resid = (m - data["y1"][bi]).abs()
    assert float(resid.max()) < 0.05, "the mean must (nearly) interpolate the observed response"


def test_materialised_route_agrees_with_the_fused_launch(data):
    """The per-function route (mgp_pairwise_dists -> mgp_kernel_apply -> mgp_perturb -> mgp_solve on
    materialised tensors, the register-resident solve kernel) and the fused launch compute the same
    posterior on a 100 k-neighbourhood slice of the full-size problem."""
    from muygpys_amd._src.gp.kernels import hip as KF
    from muygpys_amd._src.gp.muygps import hip as MF
    from muygpys_amd._src.gp.noise import hip as NF
    from muygpys_amd._src.gp.tensors import hip as TF

    nb = 100_000
    bi, ni = data["bi"][:nb], data["ni"][:nb]
    m, v = run(data, data["y1"], bi=bi, ni=ni)
    Kin = NF._homoscedastic_perturb(KF._apply(TF._pairwise_distances(data["X"], ni, "l2"), "matern15", 1.0 / 5.0), 1e-3)
    Kc = KF._apply(TF._crosswise_distances(data["X"], data["X"], bi, ni, "l2"), "matern15", 1.0 / 5.0)
    m2 = MF._muygps_posterior_mean(Kin, Kc, data["y1"][ni])
    v2 = MF._muygps_diagonal_variance(Kin, Kc, 1.0)
    assert_close(m2.cpu().numpy(), m.cpu().numpy(), 1e-3, "mean, materialised vs fused")
    assert_close(v2.cpu().numpy(), v.cpu().numpy(), 1e-3, "var, materialised vs fused")


def test_loocv_partials_of_shards_sum_to_the_whole(data):
    """mgp_loocv_*: the six fp64 partial sums of four shards (reference chunk rule) add up to those of the
    whole batch (to fp64 rounding of a different summation order), at full size."""
    from muygpys_amd import distributed as D_
    from muygpys_amd.fused import loocv_partials

    whole, mean, var = loocv_partials(data["spec"], data["X"], data["y1"], data["bi"], data["ni"])
    total = torch.zeros(6, device="cuda", dtype=torch.float64)
    means = []
    for r in range(4):
        p, mr, _ = loocv_partials(data["spec"], data["X"], data["y1"], D_.shard_rows(data["bi"], r, 4), D_.shard_rows(data["ni"], r, 4))
        total += p
        means.append(mr)
    assert torch.equal(torch.cat(means), mean), "shard outputs concatenate exactly"
    torch.testing.assert_close(total, whole, rtol=1e-11, atol=0)
    assert float(whole[3]) == float(B)
    fin = D_.finish_objective(whole.tolist(), K)
    assert 0 < fin["sigma_sq"] < 10 and np.isfinite(fin["lool"])
