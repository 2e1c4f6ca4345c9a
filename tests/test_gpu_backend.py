"""GPU parity of every hip backend function (muygpys_amd/_src/**/hip.py -> C ABI) against the
oracle and the reference-generated fixtures -- the counterpart of the reference's
tests/backend/torch_correctness.py (stage-by-stage comparison of backend functions)."""

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.conftest import spec_from_meta
from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DTYPES = ["float64", "float32"]


def _dev_inputs(g, dtype):
    td = getattr(torch, dtype)
    return to_dev(g["features"], td), to_dev(g["targets"], td), to_dev(g["batch_idx"]), to_dev(g["nn_idx"]), td


@pytest.mark.parametrize("dtype", DTYPES)
def test_tensor_family(golden, dtype):
    from muygpys_amd._src.gp import tensors as T
    from muygpys_amd._src.gp.tensors import hip as Th

    g = golden
    X, y, bi, ni, td = _dev_inputs(g, dtype)
    Xn = g["features"]
    rtol = RTOL[dtype]
    cd_ref = orc.crosswise_tensor(Xn, Xn, g["batch_idx"], g["nn_idx"])
    pd_ref = orc.pairwise_tensor(Xn, g["nn_idx"])
    cd = T._crosswise_tensor(X, X, bi, ni)
    pd = T._pairwise_tensor(X, ni)
    assert cd.shape == cd_ref.shape and pd.shape == pd_ref.shape
    assert_close(cd.cpu().numpy(), cd_ref, rtol, "crosswise diffs")
    assert_close(pd.cpu().numpy(), pd_ref, rtol, "pairwise diffs")
    for name, fn, ofn in (("F2", T._F2, orc.F2), ("l2", T._l2, orc.l2)):
        assert_close(fn(pd).cpu().numpy(), ofn(pd_ref), rtol, f"{name}(pairwise)")
        assert_close(fn(cd).cpu().numpy(), ofn(cd_ref), rtol, f"{name}(crosswise)")
        assert_close(Th._pairwise_distances(X, ni, name).cpu().numpy(), ofn(pd_ref), rtol, f"fused pairwise {name}")
        assert_close(Th._crosswise_distances(X, X, bi, ni, name).cpu().numpy(), ofn(cd_ref), rtol,
                     f"fused crosswise {name}")
    assert_close(T._batch_features_tensor(X, bi).cpu().numpy(), Xn[g["batch_idx"]], rtol, "batch features")
    assert torch.equal(T._fast_nn_update(ni).cpu(), torch.as_tensor(orc.fast_nn_update(g["nn_idx"])))
    # plain difference helpers
    pts = X[ni]
    assert_close(T._pairwise_differences(pts).cpu().numpy(), pd_ref, rtol, "pairwise_differences")
    assert_close(T._crosswise_differences(X[bi], pts).cpu().numpy(), cd_ref, rtol, "crosswise_differences")


@pytest.mark.parametrize("dtype", DTYPES)
def test_kernel_family_against_sklearn(dtype):
    """Known-answer pin the reference uses (tests/kernels.py:325-526): scikit-learn kernels."""
    from sklearn.gaussian_process.kernels import RBF, Matern

    from muygpys_amd._src.gp import kernels as K
    from muygpys_amd._src.gp import tensors as T

    rng = np.random.default_rng(11)
    Xn = rng.normal(size=(60, 4))
    td = getattr(torch, dtype)
    X = to_dev(Xn, td)
    ni = to_dev(np.arange(60)[None, :])
    ell = 1.3
    rtol = RTOL[dtype]
    F2 = T._F2(T._pairwise_tensor(X, ni))[0]
    l2 = T._l2(T._pairwise_tensor(X, ni))[0]
    assert_close(K._rbf_fn(F2 / ell**2).cpu().numpy(), RBF(length_scale=ell)(Xn), rtol, "rbf")
    for fn, nu in ((K._matern_05_fn, 0.5), (K._matern_15_fn, 1.5), (K._matern_25_fn, 2.5), (K._matern_inf_fn, np.inf)):
        assert_close(fn(l2 / ell).cpu().numpy(), Matern(length_scale=ell, nu=nu)(Xn), rtol, f"matern {nu}")
    # general smoothness (K3): the reference pins nu = 0.42 against scikit-learn (tests/kernels.py:429-526)
    for nu in (0.42, 1.0, 3.7):
        before = l2.clone()
        got = K._matern_gen_fn(l2 / ell, smoothness=nu)
        assert torch.equal(l2, before), "the hip backend does not overwrite its input"
        assert_close(got.cpu().numpy(), Matern(length_scale=ell, nu=nu)(Xn), rtol, f"matern general {nu}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_general_matern_matches_reference_fixture(dtype):
    """mgp_matern_gen_* (device Bessel function) against the reference's _matern_gen_fn
    (scipy.special.kv) on a grid of distances from 0 to 300 (tests/golden/make_golden_gen.py)."""
    from muygpys_amd._src.gp import kernels as K
    from tests.conftest import load_golden

    g = load_golden("gen_matern_function")
    td = getattr(torch, dtype)
    x = to_dev(g["dists"], td)
    for nu, want in zip(g["smoothness"], g["values"]):
        got = K._matern_gen_fn(x, smoothness=float(nu)).cpu().numpy()
        if dtype == "float64":
            np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
        else:  # the fp32 input grid differs from the fp64 one by its own rounding
            want32 = orc.matern_gen_fn(g["dists"].astype(np.float32).astype(np.float64), float(nu))
            np.testing.assert_allclose(got, want32, rtol=2e-6, atol=1e-37)


@pytest.mark.parametrize("dtype", DTYPES)
def test_noise_and_solve_family(golden, dtype):
    from muygpys_amd._src.gp import muygps as M
    from muygpys_amd._src.gp import noise as N
    from muygpys_amd._src.optimize import scale as S

    g, meta = golden, golden["meta"]
    if "Kin" not in g:
        pytest.skip("fixture stores no Kin")
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")
    td = getattr(torch, dtype)
    rtol = RTOL[dtype]
    Kin, Kc = to_dev(g["Kin"], td), to_dev(g["Kcross"], td)
    ynn = to_dev(g["batch_nn_targets"], td)
    if meta.get("hetero"):
        eps = to_dev(g["noise_table"][g["nn_idx"]], td)
        Kp = N._heteroscedastic_perturb(Kin, eps)
    else:
        Kp = N._homoscedastic_perturb(Kin, meta["noise"])
    assert_close(Kp.cpu().numpy(), g["Kin_perturbed"], rtol, "perturb")
    assert torch.equal(Kin, to_dev(g["Kin"], td)), "inputs must not be written"
    mean = M._muygps_posterior_mean(Kp, Kc, ynn)
    var = M._muygps_diagonal_variance(Kp, Kc, torch.ones((), device="cuda", dtype=td))
    assert mean.shape == g["mean"].shape
    assert_close(mean.cpu().numpy(), g["mean"], rtol, "posterior mean")
    assert_close(var.cpu().numpy(), g["var_unscaled"], rtol, "diagonal variance")
    b, k = g["nn_idx"].shape
    if meta["R"] == 1:
        s = S._analytic_scale_optim(Kp, ynn)
        assert_close(np.atleast_1d(s.cpu().numpy()), g["sigma_sq"], rtol, "sigma_sq")
        un = S._analytic_scale_optim_unnormalized(Kp, ynn)
        assert_close(np.atleast_1d(un.cpu().numpy()) / (b * k), g["sigma_sq"], rtol, "sigma_sq unnormalized")
    else:
        with pytest.raises(ValueError):
            S._analytic_scale_optim(Kp, ynn)
        un = S._analytic_scale_optim_unnormalized(Kp, ynn)
        assert_close(np.atleast_1d(un.cpu().numpy()) / (b * k), [g["sigma_sq"].sum()], rtol, "sum of sigma_sq_r")
    # fast posterior mean coefficients: K^-1 Y, and mean == Kcross . coeffs
    co = M._muygps_fast_posterior_mean_precompute(Kp, ynn)
    co_ref = orc.fast_posterior_mean_precompute(g["Kin_perturbed"], g["batch_nn_targets"])
    assert_close(co.cpu().numpy(), co_ref, 10 * rtol, "fast coefficients")
    fm = M._muygps_fast_posterior_mean(Kc, co)
    assert_close(fm.cpu().numpy(), np.squeeze(g["mean"]), 10 * rtol, "fast posterior mean")
    with pytest.raises(ValueError):
        N._homoscedastic_perturb(Kin[0], 1e-3)
    with pytest.raises(NotImplementedError):
        N._shear_perturb33(Kin, 1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_loss_family(golden, dtype):
    from muygpys_amd._src.optimize import loss as L

    g = golden
    if "lool" not in g:
        pytest.skip("multi-response fixture")
    td = getattr(torch, dtype)
    rtol = RTOL[dtype]
    mean, y, var = to_dev(g["mean"], td), to_dev(g["batch_targets"], td), to_dev(g["var_unscaled"], td)
    s = float(g["sigma_sq"][0])
    assert_close(L._mse_fn(mean, y).cpu().numpy(), g["mse"], rtol, "mse")
    assert_close(L._lool_fn(mean, y, var, s).cpu().numpy(), g["lool"], rtol, "lool")
    assert_close(L._lool_fn(mean, y, var, torch.tensor(s, device="cuda")).cpu().numpy(), g["lool"], rtol, "lool(dev)")
    assert_close(L._lool_fn_unscaled(mean, y, s * var).cpu().numpy(), g["lool"], rtol, "lool unscaled")
    assert_close(L._looph_fn(mean, y, var, s).cpu().numpy(), g["looph"], rtol, "looph")
    assert_close(L._pseudo_huber_fn(mean, y).cpu().numpy(), g["huber"], rtol, "pseudo huber")
    with pytest.raises(NotImplementedError):
        L._cross_entropy_fn(mean, y)


def test_backend_rejects_host_arrays():
    from muygpys_amd._src.gp import tensors as T

    with pytest.raises(TypeError):
        T._F2(np.zeros((3, 2)))
    with pytest.raises(TypeError):
        T._F2(torch.zeros((3, 2)))


def test_empty_inputs_through_the_new_entry_points():
    """Empty shards are legal everywhere on the path (a rank can own zero rows): the k-NN search, the
    differentiable posterior and the fused coefficient precompute return empty results."""
    from muygpys_amd.autograd import posterior
    from muygpys_amd.fused import KernelSpec, fast_coefficients
    from muygpys_amd.neighbors import NN_Wrapper

    X = torch.randn(20000, 8, device="cuda")
    y = torch.randn(20000, device="cuda")
    nbrs = NN_Wrapper(X, 10)
    idx, dist = nbrs.get_nns(X[:0])
    assert idx.shape == (0, 10) and dist.shape == (0, 10) and idx.dtype == torch.int64
    idx, dist = nbrs.get_batch_nns(torch.zeros(0, dtype=torch.int64, device="cuda"))
    assert idx.shape == (0, 10)

    x = X.clone().requires_grad_(True)
    mean, var = posterior(KernelSpec("matern15", "l2", 2.0, 1e-2), x, x,
                          torch.zeros(0, dtype=torch.int64, device="cuda"),
                          torch.zeros((0, 10), dtype=torch.int64, device="cuda"), y)
    assert mean.shape == (0,) and var.shape == (0,)
    (mean.sum() + var.sum()).backward()
    assert x.grad is not None and float(x.grad.abs().sum()) == 0.0

    coeffs, nn_fast = fast_coefficients(KernelSpec("matern15", "l2", 2.0, 1e-2), X[:0], y[:0],
                                        torch.zeros((0, 10), dtype=torch.int64, device="cuda"))
    assert coeffs.shape == (0, 10) and nn_fast.shape == (0, 10)


@pytest.mark.parametrize("responses", ["plain_tensor", "facade_table"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_family_level_lazy_route_matches_golden(golden, dtype, responses):
    """The call sequence of the reference's functor layer written against the family functions
    (what integration.install() binds): with config.state.lazy_tensors the tensor family returns
    handles, the metric / deformation / kernel / noise steps decorate them, and posterior mean,
    variance and sigma^2 come out of ONE fused launch -- on gathered responses when the response
    table is a plain tensor (the reference's own ``train_targets[nn_indices]`` materialises (b, k);
    mgp_posterior_packed_gathered_* / mgp_posterior_gathered_*), on the prepared table when it is a
    tensor of the facade (the gather stays a handle; mgp_posterior_packed_*) -- same numbers as the
    fixtures either way."""
    from muygpys_amd import lazy
    from muygpys_amd._src.gp import kernels as K
    from muygpys_amd._src.gp import muygps as M
    from muygpys_amd._src.gp import noise as N
    from muygpys_amd._src.gp import tensors as T
    from muygpys_amd._src.optimize import scale as S
    from muygpys_amd.config import config

    g, meta = golden, golden["meta"]
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")
    td = getattr(torch, dtype)
    rtol = RTOL[dtype]
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    metric = T._l2 if meta["metric"] == "l2" else T._F2
    kfn = {"rbf": K._rbf_fn, "matern05": K._matern_05_fn, "matern15": K._matern_15_fn, "matern25": K._matern_25_fn,
           "maternInf": K._matern_inf_fn}[meta["kernel"]]
    ls = meta["length_scale"]
    config.state.lazy_tensors = True
    try:
        pair, cross = T._pairwise_tensor(X, ni), T._crosswise_tensor(X, X, bi, ni)
        assert isinstance(pair, lazy.LazyDiffs) and tuple(pair.shape) == tuple(ni.shape) + (ni.shape[1], meta["d"])
        if isinstance(ls, list):  # Anisotropy.__call__: metric(diffs / l_vec), anisotropy.py:70
            lsv = to_dev(np.asarray(ls), td)
            dp, dc = metric(pair / lsv), metric(cross / lsv)
        else:                     # Isotropy: metric at construction, then x / l or x / l^2, metric.py:241,264
            div = ls if meta["metric"] == "l2" else ls**2
            dp, dc = metric(pair) / div, metric(cross) / div
        Kin, Kc = kfn(dp), kfn(dc)
        assert isinstance(Kin, lazy.LazyCov) and tuple(Kin.shape) == tuple(ni.shape) + (ni.shape[1],)
        if responses == "facade_table":
            from muygpys_amd import integration

            ynn = integration.table(y)[ni]  # gp/muygps.py:474 on a tensor of the facade
            assert isinstance(ynn, lazy.LazyTargets) and tuple(ynn.shape) == tuple(ni.shape) + tuple(y.shape[1:])
        else:
            ynn = y[ni]
            assert isinstance(ynn, torch.Tensor)
        if meta.get("hetero"):
            Kp = N._heteroscedastic_perturb(Kin, T._make_heteroscedastic_tensor(to_dev(g["noise_table"], td), ni))
        else:
            Kp = N._homoscedastic_perturb(Kin, meta["noise"])
        mean = M._muygps_posterior_mean(Kp, Kc, ynn)
        var = M._muygps_diagonal_variance(Kp, Kc, 1.0)
        assert len(Kin.cache["entries"]) == 1, "mean and variance share one launch"
        assert_close(mean.cpu().numpy().reshape(g["mean"].shape), g["mean"], rtol, "mean")
        assert_close(var.cpu().numpy(), g["var_unscaled"], rtol, "var")
        if meta["R"] == 1:
            sig = S._analytic_scale_optim(Kp, ynn)
            assert len(Kin.cache["entries"]) == 1, "sigma^2 comes from the same launch"
            assert_close(sig.cpu().numpy().reshape(-1), g["sigma_sq"], rtol, "sigma_sq")
        # a handle is still a tensor on demand
        if "Kin" in g:
            assert_close(torch.as_tensor(Kin.materialize()).cpu().numpy(), g["Kin"], rtol, "Kin materialised")
    finally:
        config.state.lazy_tensors = False
