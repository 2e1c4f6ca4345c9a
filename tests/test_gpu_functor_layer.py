"""GPU: the functor layer with its default (hip) backends, through both routes --
materialised per-function kernels and lazy handles -> one fused launch -- against the
reference-generated fixtures.  Mirrors the reference's cross-backend test
(tests/backend/torch_correctness.py:154-331,1263-1490)."""

import numpy as np
import pytest

from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

NU = {"matern05": 0.5, "matern15": 1.5, "matern25": 2.5, "maternInf": np.inf}


def hip_model(meta, g, td, bounds=None, noise_bounds="fixed"):
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import F2, Anisotropy, Isotropy, l2
    from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter, VectorParameter
    from muygpys_amd.gp.kernels import RBF, Matern
    from muygpys_amd.gp.noise import HeteroscedasticNoise, HomoscedasticNoise
    from muygpys_amd.gp.tensors import make_heteroscedastic_tensor

    bounds = bounds or {}

    def P(name, val):
        return Parameter(val, bounds[name]) if name in bounds else Parameter(val)

    metric = l2 if meta["metric"] == "l2" else F2
    ls = meta["length_scale"]
    if isinstance(ls, list):
        deformation = Anisotropy(metric, VectorParameter(*[P(f"length_scale{i}", v) for i, v in enumerate(ls)]))
    else:
        deformation = Isotropy(metric, P("length_scale", ls))
    kernel = RBF(deformation=deformation) if meta["kernel"] == "rbf" else Matern(
        smoothness=Parameter(NU[meta["kernel"]]), deformation=deformation
    )
    if meta.get("hetero"):
        noise = HeteroscedasticNoise(make_heteroscedastic_tensor(to_dev(g["noise_table"], td), to_dev(g["nn_idx"])))
    else:
        noise = HomoscedasticNoise(meta["noise"], noise_bounds)
    return MuyGPS(kernel=kernel, noise=noise, scale=AnalyticScale())


def _skip_ill_conditioned(meta, dtype):
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("materialize", [True, False])
def test_call_sequence(golden, dtype, materialize):
    from muygpys_amd import lazy

    g, meta = golden, golden["meta"]
    _skip_ill_conditioned(meta, dtype)
    td = getattr(torch, dtype)
    rtol = RTOL[dtype]
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    m = hip_model(meta, g, td)
    cross, pair, y_b, y_nn = m.make_train_tensors(bi, ni, X, y, materialize=materialize)
    assert tuple(cross.shape) == (g["crosswise"].shape if "crosswise" in g else cross.shape)
    assert lazy.is_lazy(pair) == (not materialize)
    if "pairwise" in g:
        assert_close(lazy.force(pair).cpu().numpy(), g["pairwise"], rtol, "pairwise")
        assert_close(lazy.force(cross).cpu().numpy(), g["crosswise"], rtol, "crosswise")
    assert_close(y_b.cpu().numpy(), g["batch_targets"], rtol, "batch targets")
    assert_close(lazy.force(y_nn).cpu().numpy(), g["batch_nn_targets"], rtol, "nn targets")
    Kin, Kc = m.kernel(pair), m.kernel(cross)
    assert lazy.is_lazy(Kin) == (not materialize)
    assert_close(lazy.force(Kc).cpu().numpy(), g["Kcross"], rtol, "Kcross")
    if "Kin" in g:
        assert_close(lazy.force(Kin).cpu().numpy(), g["Kin"], rtol, "Kin")
        assert_close(lazy.force(m.noise.perturb(Kin)).cpu().numpy(), g["Kin_perturbed"], rtol, "Kin perturbed")
    mean = m.posterior_mean(Kin, Kc, y_nn)
    var0 = m.get_opt_var_fn()(Kin, Kc)
    assert mean.shape == g["mean"].shape
    assert_close(mean.cpu().numpy(), g["mean"], rtol, "mean")
    assert_close(var0.cpu().numpy(), g["var_unscaled"], rtol, "unscaled variance")
    if meta["R"] == 1:
        m = m.optimize_scale(pair, y_nn)
        assert m.scale.trained
        assert_close([m.scale()], g["sigma_sq"], rtol, "sigma_sq")
        assert_close(m.posterior_variance(Kin, Kc).cpu().numpy(), g["var_scaled"], rtol, "scaled variance")
    if not materialize:
        # mean, variance (and the scale inside an objective) share ONE fused launch
        assert len(Kin.cache) == 1


@pytest.mark.parametrize("materialize", [True, False])
def test_objective_probes(golden, materialize):
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from muygpys_amd.optimize.loss import looph_fn, lool_fn, mse_fn, pseudo_huber_fn

    g, meta = golden, golden["meta"]
    if "probe_values" not in g:
        pytest.skip("no probes in this fixture")
    td = torch.float64
    probes = meta["probes"]
    bounds = {k: (1e-6, 1e6) for p in probes for k in p if k != "noise"}
    noise_bounds = (1e-8, 1e2) if any("noise" in p for p in probes) and not meta.get("hetero") else "fixed"
    m = hip_model(meta, g, td, bounds=bounds, noise_bounds=noise_bounds)
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(g["batch_idx"]), to_dev(g["nn_idx"]), X, y, materialize=materialize)
    for row, lfn in enumerate((lool_fn, mse_fn, looph_fn, pseudo_huber_fn)):
        obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=lfn)
        vals = [float(obj(**p)) for p in probes]
        assert_close(vals, g["probe_values"][row], 1e-5, f"objective row {row}")


def test_lbfgsb_on_gpu_improves_objective():
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from tests.conftest import load_golden

    g = load_golden("m15_iso_knn_k30_d40_c2")
    meta = dict(g["meta"])
    meta["length_scale"] = 1.5
    td = torch.float64
    m = hip_model(meta, g, td, bounds={"length_scale": (0.5, 20.0)})
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(g["batch_idx"]), to_dev(g["nn_idx"]), X, y)
    obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair)
    new = L_BFGS_B_optimize(m, y_b, y_nn, cross, pair)
    ls = new.kernel.deformation.length_scale()
    assert 0.5 <= ls <= 20.0 and float(obj(length_scale=ls)) >= float(obj(length_scale=1.5))


def test_singular_neighbourhood_raises_linalgerror():
    """Reference behaviour: linalg.solve raises numpy.linalg.LinAlgError on a singular system."""
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import F2, Isotropy
    from muygpys_amd.gp.hyperparameter import Parameter
    from muygpys_amd.gp.kernels import RBF
    from muygpys_amd.gp.noise import HomoscedasticNoise

    X = torch.randn(50, 4, device="cuda", dtype=torch.float64)
    y = torch.randn(50, device="cuda", dtype=torch.float64)
    ni = torch.tensor([[1, 1, 2, 3], [4, 5, 6, 7]], device="cuda")  # duplicate neighbour, zero nugget
    bi = torch.tensor([0, 8], device="cuda")
    m = MuyGPS(kernel=RBF(deformation=Isotropy(F2, length_scale=Parameter(1.0))), noise=HomoscedasticNoise(0.0))
    for materialize in (False, True):
        cross, pair, y_nn = m.make_predict_tensors(bi, ni, None, X, y, materialize=materialize)
        Kin, Kc = m.kernel(pair), m.kernel(cross)
        with pytest.raises(np.linalg.LinAlgError):
            m.posterior_mean(Kin, Kc, y_nn)


def test_deferred_spd_check_raises_one_call_late_or_at_the_flush():
    """config.state.check_spd = "deferred": the counter of a launch is looked at when the next checked call arrives
    (or at _lib.flush_spd_checks()), so a loop of evaluations does not wait for every launch; the singular
    neighbourhood's outputs are NaN meanwhile and the same LinAlgError is raised, one call late."""
    from muygpys_amd import _lib
    from muygpys_amd.config import config
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import F2, Isotropy
    from muygpys_amd.gp.hyperparameter import Parameter
    from muygpys_amd.gp.kernels import RBF
    from muygpys_amd.gp.noise import HomoscedasticNoise

    g = torch.Generator().manual_seed(21)
    X = torch.randn(50, 4, generator=g, dtype=torch.float64).cuda()
    y = torch.randn(50, generator=g, dtype=torch.float64).cuda()
    bad_ni = torch.tensor([[1, 1, 2, 3], [4, 5, 6, 7]], device="cuda")  # duplicate neighbour, zero nugget
    good_ni = torch.tensor([[1, 9, 2, 3], [4, 5, 6, 7]], device="cuda")
    bi = torch.tensor([0, 8], device="cuda")
    m = MuyGPS(kernel=RBF(deformation=Isotropy(F2, length_scale=Parameter(1.0))), noise=HomoscedasticNoise(0.0))

    def mean_of(ni):
        cross, pair, y_nn = m.make_predict_tensors(bi, ni, None, X, y)
        return m.posterior_mean(m.kernel(pair), m.kernel(cross), y_nn)

    before = config.state.check_spd
    config.state.check_spd = "deferred"
    try:
        out = mean_of(bad_ni)                      # no error yet ...
        assert bool(torch.isnan(torch.as_tensor(out)[0]).all()) and bool(torch.isfinite(torch.as_tensor(out)[1]).all())
        with pytest.raises(np.linalg.LinAlgError, match="one call late"):
            mean_of(good_ni)                       # ... the next checked call reports it
        _lib.flush_spd_checks()                    # (the good call's own counter: clean)
        mean_of(bad_ni)
        with pytest.raises(np.linalg.LinAlgError):
            _lib.flush_spd_checks()
        _lib.flush_spd_checks()                    # nothing pending: no error
    finally:
        config.state.check_spd = before
        _lib._SPD_PENDING.clear()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("materialize", [False, True])
def test_free_smoothness_model_matches_reference(dtype, materialize):
    """K3 through the functor layer: a Matern whose smoothness is a free hyper-parameter selects the
    general Bessel form (gp/kernels/matern.py:61-81); kernel tensors, posterior, sigma^2 and LOOCV
    objective values at smoothness probes against the reference (tests/golden/make_golden_gen.py),
    from lazy handles and from materialised tensors."""
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from muygpys_amd.optimize.loss import lool_fn, mse_fn
    from tests.conftest import load_golden

    g = load_golden("gen_m042_iso_k10_d6")
    meta = g["meta"]
    td = getattr(torch, dtype)
    rtol = RTOL[dtype]
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    m = MuyGPS(kernel=Matern(smoothness=Parameter(meta["smoothness"], (0.1, 5.0)),
                             deformation=Isotropy(l2, length_scale=Parameter(meta["length_scale"]))),
               noise=HomoscedasticNoise(meta["noise"]), scale=AnalyticScale())
    cross, pair, y_b, y_nn = m.make_train_tensors(bi, ni, X, y, materialize=materialize)
    Kin, Kc = m.kernel(pair), m.kernel(cross)
    assert_close(Kin.cpu().numpy(), g["Kin"], rtol, "Kin")
    assert_close(Kc.cpu().numpy(), g["Kcross"], rtol, "Kcross")
    assert_close(m.posterior_mean(Kin, Kc, y_nn).cpu().numpy(), g["mean"], rtol, "mean")
    assert_close(m.get_opt_var_fn()(Kin, Kc).cpu().numpy(), g["var_unscaled"], rtol, "var")
    m = m.optimize_scale(pair, y_nn)
    assert_close(np.asarray(float(m.scale())).reshape(-1), g["sigma_sq"], rtol, "sigma_sq")
    for loss, want in ((lool_fn, g["probe_lool"]), (mse_fn, g["probe_mse"])):
        obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=loss)
        got = [float(obj(smoothness=p)) for p in meta["probes"]]
        assert_close(got, want, rtol, "objective at smoothness probes")


@pytest.mark.parametrize("name", ["m15_iso_l2_k10_d8", "rbf_iso_F2_k10_d1", "m25_iso_l2_k30_d40"])
@pytest.mark.parametrize("lazy", [True, False])
def test_difference_isotropy_is_isotropy_on_differences(name, lazy):
    """DifferenceIsotropy (gp/deformation/isotropy.py:165-276): ``metric(diffs / l)`` on difference tensors.
    For the stock metrics it is the same function of the points as Isotropy, so an isotropic fixture must be
    reproduced -- through the lazy route (one fused launch) and on materialised (b, k, k, d) differences."""
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import F2, DifferenceIsotropy, l2
    from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
    from muygpys_amd.gp.kernels import RBF, Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from tests.conftest import load_golden

    g = load_golden(name)
    meta = g["meta"]
    td = torch.float64
    deformation = DifferenceIsotropy(l2 if meta["metric"] == "l2" else F2, Parameter(meta["length_scale"]))
    kernel = RBF(deformation=deformation) if meta["kernel"] == "rbf" else Matern(
        smoothness=Parameter(NU[meta["kernel"]]), deformation=deformation)
    m = MuyGPS(kernel=kernel, noise=HomoscedasticNoise(meta["noise"]), scale=AnalyticScale())
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    cross, pair, y_b, y_nn = m.make_train_tensors(bi, ni, X, y, materialize=not lazy)
    assert tuple(pair.shape) == tuple(ni.shape) + (ni.shape[1], meta["d"])  # differences, not distances
    Kin, Kc = m.kernel(pair), m.kernel(cross)
    mean = m.posterior_mean(Kin, Kc, y_nn)
    var = m.get_opt_var_fn()(Kin, Kc)
    assert_close(torch.as_tensor(mean).cpu().numpy().reshape(g["mean"].shape), g["mean"], 1e-5, "mean")
    assert_close(torch.as_tensor(var).cpu().numpy(), g["var_unscaled"], 1e-5, "var")
