"""GPU: the fast-posterior-mean workflow (coefficient precompute + fused prediction kernel)
against the oracle's restatement of the reference (examples/fast_posterior_mean.py:317-400,
_src/gp/muygps/numpy.py:70-95; reference test: tests/backend/torch_correctness.py:786-1165)."""

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def oracle_fast(X, y, Q, k, spec):
    from sklearn.neighbors import NearestNeighbors

    nbrs = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(X)
    nn = nbrs.kneighbors(X, return_distance=False)[:, 1:]
    nn_fast = orc.fast_nn_update(nn)
    pd = orc.pairwise_tensor(X, nn_fast)
    _, Kin = orc.kernel_tensors(spec, pd[:, 0], pd)
    coeffs = orc.fast_posterior_mean_precompute(orc.perturb(spec, Kin, nn_fast), y[nn_fast])
    closest = nbrs.kneighbors(Q, n_neighbors=1, return_distance=False)[:, 0]
    cset = nn_fast[closest]
    cd = orc.crosswise_tensor(Q, X, np.arange(len(Q)), cset)
    Kc, _ = orc.kernel_tensors(spec, cd, pd[:1])
    return nn, closest, cset, coeffs, orc.fast_posterior_mean(Kc, coeffs[closest])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", [("matern15", "l2", 2.5, 8, 10, 1), ("rbf", "F2", 4.0, 40, 30, 1),
                                  ("matern25", "l2", [1.0, 2.0, 0.7, 1.5, 3.0], 5, 12, 3)])
def test_fast_workflow(dtype, case):
    from muygpys_amd.fused import KernelSpec, fast_coefficients, fast_posterior_mean
    from muygpys_amd.neighbors import NN_Wrapper

    kernel, metric, ls, d, k, R = case
    rng = np.random.default_rng(17)
    X = rng.normal(size=(1500, d))
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.05 * rng.normal(size=(1500, R))
    y = Y[:, 0] if R == 1 else Y
    Q = rng.normal(size=(300, d))
    ospec = orc.Spec(kernel, metric, np.asarray(ls) if isinstance(ls, list) else ls, 1e-2)
    nn, closest, cset, coeffs_ref, mean_ref = oracle_fast(X, y, Q, k, ospec)
    td = getattr(torch, dtype)
    Xd, yd, Qd = to_dev(X, td), to_dev(y, td), to_dev(Q, td)
    spec = KernelSpec(kernel, metric, ls, 1e-2)
    nbrs = NN_Wrapper(Xd, k)
    nn_d, _ = nbrs.get_batch_nns(torch.arange(1500, device="cuda"))
    assert np.array_equal(nn_d.cpu().numpy(), nn)
    coeffs, nn_fast = fast_coefficients(spec, Xd, yd, nn_d, chunk=400)
    rtol = 3 * RTOL[dtype]
    assert_close(coeffs.cpu().numpy(), coeffs_ref, rtol, "coefficients")
    closest_d = nbrs.get_nns(Qd)[0][:, 0]
    assert np.array_equal(closest_d.cpu().numpy(), closest)
    mean = fast_posterior_mean(spec, Qd, Xd, None, nn_fast[closest_d], coeffs, closest_d)
    assert mean.shape == mean_ref.shape
    assert_close(mean.cpu().numpy(), mean_ref, rtol, "fast posterior mean")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("name", __import__("tests.conftest", fromlist=["fast_golden_names"]).fast_golden_names())
def test_fast_kernels_match_reference_fixture(name, dtype):
    """The fused coefficient precompute, the materialising precompute and the fused prediction
    kernel against the reference's own fast-posterior-mean outputs (tests/golden/make_golden_fast.py)."""
    from muygpys_amd.fused import KernelSpec, fast_coefficients, fast_posterior_mean
    from tests.conftest import load_golden

    g = load_golden(name)
    meta = g["meta"]
    td = getattr(torch, dtype)
    spec = KernelSpec(meta["kernel"], meta["metric"], meta["length_scale"], meta["noise"])
    Xd, yd, Qd = to_dev(g["features"], td), to_dev(g["targets"], td), to_dev(g["test_features"], td)
    nn = to_dev(g["train_nn"])
    rtol = 3 * RTOL[dtype]
    for fused in (True, False):
        coeffs, nn_fast = fast_coefficients(spec, Xd, yd, nn, fused=fused)
        assert np.array_equal(nn_fast.cpu().numpy(), g["train_nn_fast"])
        assert_close(coeffs.cpu().numpy(), g["coeffs"], rtol, f"coefficients (fused={fused})")
    closest = to_dev(g["closest_neighbor"])
    mean = fast_posterior_mean(spec, Qd, Xd, None, nn_fast[closest], coeffs, closest)
    assert_close(mean.cpu().numpy(), g["fast_mean"], rtol, "fast posterior mean")


def test_fused_coefficients_match_the_materialising_path_at_scale():
    """mgp_fast_coefficients_f32 (one launch, every workgroup looping over many tasks) against the
    per-function path (pairwise distances -> kernel -> nugget -> mgp_solve) in fp64."""
    from muygpys_amd.fused import KernelSpec, fast_coefficients

    g = torch.Generator().manual_seed(21)
    n, d, k = 150_000, 40, 30
    X = torch.randn(n, d, generator=g)
    y = torch.sin(X @ (torch.randn(d, generator=g) / d**0.5)) + 0.05 * torch.randn(n, generator=g)
    nn = torch.randint(0, n, (n, k), generator=g)
    spec = KernelSpec("matern15", "l2", 6.0, 1e-2)
    c32, nnf = fast_coefficients(spec, X.cuda(), y.cuda(), nn.cuda())
    c64, nnf64 = fast_coefficients(spec, X.double().cuda(), y.double().cuda(), nn.cuda(), fused=False)
    assert torch.equal(nnf, nnf64) and c32.dtype == torch.float32 and c32.shape == (n, k)
    assert_close(c32.cpu().numpy(), c64.cpu().numpy(), 3 * RTOL["float32"], "coefficients")
    f64, _ = fast_coefficients(spec, X.double().cuda(), y.double().cuda(), nn.cuda())
    assert_close(f64.cpu().numpy(), c64.cpu().numpy(), RTOL["float64"], "fused fp64 coefficients")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fused_coefficients_wide_neighbourhoods(dtype):
    """k = 50 (64-slot kernel, one neighbourhood per wave), anisotropic, heteroscedastic table."""
    from muygpys_amd.fused import KernelSpec, fast_coefficients

    td = getattr(torch, dtype)
    g = torch.Generator().manual_seed(8)
    n, d, k = 30_000, 8, 50
    X = torch.randn(n, d, generator=g)
    y = torch.sin(X.sum(1)) + 0.05 * torch.randn(n, generator=g)
    nn = torch.randint(0, n, (n, k), generator=g)
    noise = (10.0 ** (-3 + 2 * torch.rand(n, generator=g))).to(td).cuda()
    spec = KernelSpec("matern25", "l2", [1.5, 2.0, 1.0, 3.0, 2.5, 1.2, 1.8, 2.2], noise)
    got, _ = fast_coefficients(spec, X.to(td).cuda(), y.to(td).cuda(), nn.cuda())
    spec64 = KernelSpec("matern25", "l2", [1.5, 2.0, 1.0, 3.0, 2.5, 1.2, 1.8, 2.2], noise.double())
    ref, _ = fast_coefficients(spec64, X.double().cuda(), y.double().cuda(), nn.cuda(), fused=False)
    assert_close(got.cpu().numpy(), ref.cpu().numpy(), RTOL[dtype] * (3 if dtype == "float32" else 1), "coefficients")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("name", __import__("tests.conftest", fromlist=["fast_golden_names"]).fast_golden_names())
def test_family_level_fast_mean_on_lazy_handles_is_one_fused_launch(name, dtype, monkeypatch):
    """The reference's own call sequence (examples/from_indices.py:93-118: crosswise_tensor -> kernel ->
    muygps.fast_posterior_mean(Kcross, coeffs[closest_index])) on lazy handles: the family function
    ``_muygps_fast_posterior_mean`` ends in the fused prediction kernel (no (b, k) covariance tensor, no einsum) and
    reproduces the reference's fast-mean output."""
    from muygpys_amd import fused as F
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Anisotropy, Isotropy, F2, l2
    from muygpys_amd.gp.hyperparameter import Parameter, VectorParameter
    from muygpys_amd.gp.kernels import RBF, Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from tests.conftest import load_golden

    g = load_golden(name)
    meta = g["meta"]
    td = getattr(torch, dtype)
    Xd, Qd = to_dev(g["features"], td), to_dev(g["test_features"], td)
    metric = l2 if meta["metric"] == "l2" else F2
    ls = meta["length_scale"]
    deformation = (Anisotropy(metric, VectorParameter(*[Parameter(v) for v in ls])) if isinstance(ls, list)
                   else Isotropy(metric, Parameter(ls)))
    if meta["kernel"] == "rbf":
        kernel = RBF(deformation=deformation)
    else:
        kernel = Matern(smoothness=Parameter({"matern05": 0.5, "matern15": 1.5, "matern25": 2.5}[meta["kernel"]]),
                        deformation=deformation)
    m = MuyGPS(kernel=kernel, noise=HomoscedasticNoise(meta["noise"]))
    calls = []
    real = F.fast_posterior_mean
    monkeypatch.setattr(F, "fast_posterior_mean", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    closest = to_dev(g["closest_neighbor"])
    cset = to_dev(g["closest_set"])
    coeffs = to_dev(g["coeffs"], td)
    crosswise = m.kernel.deformation.crosswise_tensor(Qd, Xd, torch.arange(Qd.shape[0], device="cuda"), cset, lazy=True)
    Kcross = m.kernel(crosswise)
    mean = m.fast_posterior_mean(Kcross, coeffs[closest])
    torch.cuda.synchronize()
    assert calls == [1], "the lazy crosswise handle must end in the fused prediction kernel"
    assert_close(mean.cpu().numpy(), g["fast_mean"], 3 * RTOL[dtype], "fast posterior mean (functor layer)")
    # the materialised route (the reference's tensors) agrees
    dense = m.fast_posterior_mean(Kcross.materialize(), coeffs[closest])
    assert_close(dense.cpu().numpy(), g["fast_mean"], 3 * RTOL[dtype], "fast posterior mean (materialised)")


@pytest.mark.parametrize("k", [63, 64, 70])
def test_family_level_fast_mean_on_lazy_handles_beyond_a_wavefront(k):
    """nn_count + 1 > 64 is more than mgp_fast_mean.hip serves (one test point per 32 / 64 lanes): the family function
    must fall back to the materialised covariance + the reference's einsum (_src/gp/muygps/numpy.py:70-77) instead of
    raising; k = 63 still takes the fused kernel."""
    from muygpys_amd._src.gp.muygps.hip import _muygps_fast_posterior_mean
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import Parameter
    from muygpys_amd.gp.kernels import Matern

    rng = np.random.default_rng(300 + k)
    n, b, d = 400, 57, 8
    X, Q = rng.normal(size=(n, d)), rng.normal(size=(b, d))
    cset = np.stack([rng.choice(n, size=k, replace=False) for _ in range(b)]).astype(np.int64)
    co = rng.normal(size=(b, k))
    kernel = Matern(smoothness=Parameter(1.5), deformation=Isotropy(l2, Parameter(2.0)))
    Xd, Qd = to_dev(X, torch.float64), to_dev(Q, torch.float64)
    crosswise = kernel.deformation.crosswise_tensor(Qd, Xd, torch.arange(b, device="cuda"), to_dev(cset), lazy=True)
    Kcross = kernel(crosswise)
    got = _muygps_fast_posterior_mean(Kcross, to_dev(co, torch.float64))
    torch.cuda.synchronize()
    r = np.sqrt(((Q[:, None, :] - X[cset]) ** 2).sum(-1)) / 2.0
    Kc = (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r)
    assert_close(got.cpu().numpy(), (Kc * co).sum(1), RTOL["float64"], f"fast mean, k = {k}")
