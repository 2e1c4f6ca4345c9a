"""Analytic gradients of the LOOCV objective for L-BFGS-B (round 5): one forward evaluation and one
backward launch (mgp_loocv_backward_*: the vector-Jacobian kernel with the cotangent of y^T K^-1 y chained in) instead of
p + 1 finite-difference evaluations per iteration (reference: _src/optimize/chassis/numpy.py:57-81, which has no analytic
gradient; torch autograd over torch/muygps_layer.py:129-164 is how the reference differentiates this path).

Pinned: the value is the functor layer's own objective (which is pinned to the reference's probe values), the gradient
agrees with central finite differences of that objective, the optimum equals the finite-difference driver's, in a
fraction of the fused launches."""

import numpy as np
import pytest

from tests.util import to_dev

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a ROCm device")]


def _model(kernel, aniso, d, ls0, noise, scale=None):
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Anisotropy, F2, Isotropy, l2
    from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter, VectorParameter
    from muygpys_amd.gp.kernels import RBF, Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise

    metric = F2 if kernel == "rbf" else l2
    if aniso:
        deformation = Anisotropy(metric, VectorParameter(*[Parameter(float(v), (0.2, 20.0)) for v in ls0]))
    else:
        deformation = Isotropy(metric, Parameter(float(ls0), (0.2, 20.0)))
    k = RBF(deformation=deformation) if kernel == "rbf" else Matern(
        smoothness=Parameter({"matern05": 0.5, "matern15": 1.5, "matern25": 2.5}[kernel]), deformation=deformation)
    return MuyGPS(kernel=k, noise=HomoscedasticNoise(noise), scale=AnalyticScale() if scale is None else scale)


def _data(seed, n, b, k, d, planted):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d))
    y = np.sin((X / planted) @ rng.normal(size=d) / np.sqrt(d)) + 0.05 * rng.normal(size=n)
    bi = rng.choice(n, size=b, replace=False)
    # true nearest neighbours in the planted metric (self excluded): an objective with a real optimum
    Z = X / planted
    d2 = ((Z[bi, None, :] - Z[None, :, :]) ** 2).sum(-1)
    d2[np.arange(b), bi] = np.inf
    ni = np.argsort(d2, axis=1)[:, :k]
    return X, y, bi, ni


@pytest.mark.parametrize("kernel,aniso,k,d", [("matern15", False, 12, 6), ("matern25", True, 30, 8), ("rbf", True, 20, 4),
                                              ("matern15", True, 50, 8)])
@pytest.mark.parametrize("loss", ["lool", "mse"])
def test_value_is_the_objective_and_gradient_matches_finite_differences(kernel, aniso, k, d, loss):
    from muygpys_amd._src.optimize.chassis.hip import _analytic_value_and_grad
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from muygpys_amd.optimize.loss import lool_fn, mse_fn

    planted = np.linspace(0.7, 1.6, d) if aniso else np.full(d, 1.2)
    X, y, bi, ni = _data(21 + k, 1500, 400, k, d, planted)
    ls0 = np.linspace(1.1, 1.9, d) if aniso else 1.4
    m = _model(kernel, aniso, d, ls0, 1e-2)
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    lfn = lool_fn if loss == "lool" else mse_fn
    obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=lfn)
    names, x0, _ = m.get_opt_params()
    vg = _analytic_value_and_grad(m, obj, names)
    value, grad = vg(np.asarray(x0, dtype=np.float64))
    f = lambda x: -float(obj(**{n_: float(v) for n_, v in zip(names, x)}))  # noqa: E731  (the minimised function)
    np.testing.assert_allclose(value, f(x0), rtol=1e-10)
    fd = np.zeros(len(x0))
    for j in range(len(x0)):
        h = 1e-5 * max(1.0, abs(x0[j]))
        xp, xm = np.array(x0, dtype=np.float64), np.array(x0, dtype=np.float64)
        xp[j] += h
        xm[j] -= h
        fd[j] = (f(xp) - f(xm)) / (2 * h)
    np.testing.assert_allclose(grad, fd, rtol=2e-6, atol=2e-7 * np.abs(fd).max())


@pytest.mark.parametrize("scale_kind", ["fixed", "plain", "analytic3"])
def test_gradient_follows_the_scale_the_objective_was_built_with(scale_kind):
    """lool divides by whatever ``muygps.scale.get_opt_fn`` returns (reference: gp/hyperparameter/scale.py:60-63 --
    the constant ``muygps.scale()`` for FixedScale / ScaleFn -- and :172-219 -- the closed form, optionally iterated).
    The analytic route must differentiate THAT function: value equal to the objective's, gradient equal to its central
    differences, for each of them (round-5 advice: it always assumed the one-pass closed form)."""
    from muygpys_amd._src.optimize.chassis.hip import _analytic_value_and_grad
    from muygpys_amd.gp.hyperparameter import AnalyticScale, FixedScale
    from muygpys_amd.gp.hyperparameter.scale import ScaleFn
    from muygpys_amd.optimize import L_BFGS_B_optimize

    d, k = 5, 16
    scale = {"fixed": FixedScale(val=0.37), "plain": ScaleFn(val=2.5), "analytic3": AnalyticScale(iteration_count=3, val=0.8)}[scale_kind]
    X, y, bi, ni = _data(77, 1200, 300, k, d, np.linspace(0.8, 1.5, d))
    m = _model("matern25", True, d, np.linspace(1.0, 1.8, d), 1e-2, scale=scale)
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair)
    names, x0, _ = m.get_opt_params()
    value, grad = _analytic_value_and_grad(m, obj, names)(np.asarray(x0, dtype=np.float64))
    f = lambda x: -float(obj(**{n_: float(v) for n_, v in zip(names, x)}))  # noqa: E731
    np.testing.assert_allclose(value, f(x0), rtol=1e-10)
    fd = np.zeros(len(x0))
    for j in range(len(x0)):
        h = 1e-5 * max(1.0, abs(x0[j]))
        xp, xm = np.array(x0, dtype=np.float64), np.array(x0, dtype=np.float64)
        xp[j] += h
        xm[j] -= h
        fd[j] = (f(xp) - f(xm)) / (2 * h)
    np.testing.assert_allclose(grad, fd, rtol=2e-6, atol=2e-7 * np.abs(fd).max())
    if scale_kind != "analytic3":
        # ... and it is NOT the analytic-scale function's gradient (what round 5 returned whatever the scale)
        m_an = _model("matern25", True, d, np.linspace(1.0, 1.8, d), 1e-2)
        cross2, pair2, y_b2, y_nn2 = m_an.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
        obj_an = L_BFGS_B_optimize.make_obj_fn(m_an, y_b2, y_nn2, cross2, pair2)
        _, grad_an = _analytic_value_and_grad(m_an, obj_an, names)(np.asarray(x0, dtype=np.float64))
        assert np.abs(grad_an - grad).max() > 1e-3 * np.abs(grad).max()


def test_a_foreign_scale_fn_is_refused():
    from muygpys_amd._src.optimize.chassis.hip import _analytic_value_and_grad
    from muygpys_amd.optimize.loss import lool_fn
    from muygpys_amd.optimize.objective import make_loo_crossval_fn

    X, y, bi, ni = _data(9, 800, 100, 10, 4, np.ones(4))
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    m = _model("matern15", False, 4, 1.3, 1e-2)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    obj = make_loo_crossval_fn(lool_fn, m.kernel.get_opt_fn(), m.get_opt_mean_fn(), m.get_opt_var_fn(),
                               lambda Kin, nn_targets, **kw: 1.0, pair, cross, y_nn, y_b)
    with pytest.raises(ValueError, match="scale_fn"):
        _analytic_value_and_grad(m, obj, m.get_opt_params()[0])


def test_lbfgsb_with_analytic_gradients_reaches_the_finite_difference_optimum_in_fewer_launches(monkeypatch):
    """Config 4 in miniature (anisotropic Matern-3/2, fp64, k = 50, d = 8): both drivers to the same optimum (1e-4
    relative on every length scale), the analytic one in at most a quarter of the fused forward launches."""
    from muygpys_amd import fused as F
    from muygpys_amd.optimize import L_BFGS_B_optimize

    d, k = 8, 50
    planted = np.array([0.6, 0.8, 1.0, 1.2, 1.4, 0.9, 1.1, 1.5])
    X, y, bi, ni = _data(4, 6000, 1500, k, d, planted)
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    launches = {"fwd": 0}
    real_post, real_loocv = F.posterior_mean_var, F.loocv_partials
    monkeypatch.setattr(F, "posterior_mean_var", lambda *a, **kw: (launches.__setitem__("fwd", launches["fwd"] + 1), real_post(*a, **kw))[1])
    monkeypatch.setattr(F, "loocv_partials", lambda *a, **kw: (launches.__setitem__("fwd", launches["fwd"] + 1), real_loocv(*a, **kw))[1])
    import muygpys_amd.lazy_eval as LE  # (the functor layer's launch site imports the function by name at call time)

    results = {}
    for analytic in (False, True):
        m = _model("matern15", True, d, np.full(d, 2.0), 1e-3)
        cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
        launches["fwd"] = 0
        new = L_BFGS_B_optimize(m, y_b, y_nn, cross, pair, analytic_gradient=analytic, options={"ftol": 1e-14, "gtol": 1e-9})
        results[analytic] = (np.array([float(v) for v in new.kernel.deformation.length_scale()]), launches["fwd"])
    ls_fd, n_fd = results[False]
    ls_an, n_an = results[True]
    np.testing.assert_allclose(ls_an, ls_fd, rtol=1e-4)
    assert n_fd > 0 and n_an > 0 and n_an * 4 <= n_fd, (n_an, n_fd)
    # ... and it is an optimum of THIS objective: closer to the planted scales (up to their common factor) than the start
    rel = lambda v: np.median(((v / v.mean()) / (planted / planted.mean()) - 1.0) ** 2)  # noqa: E731
    assert rel(ls_an) < 0.05, (ls_an, planted)


def _fd_check(m, obj, rtol=2e-6):
    from muygpys_amd._src.optimize.chassis.hip import _analytic_value_and_grad

    names, x0, _ = m.get_opt_params()
    value, grad = _analytic_value_and_grad(m, obj, names)(np.asarray(x0, dtype=np.float64))
    f = lambda x: -float(obj(**{n_: float(v) for n_, v in zip(names, x)}))  # noqa: E731  (the minimised function)
    np.testing.assert_allclose(value, f(x0), rtol=1e-10)
    fd = np.zeros(len(x0))
    for j in range(len(x0)):
        h = 1e-5 * max(1e-2, abs(x0[j]))
        xp, xm = np.array(x0, dtype=np.float64), np.array(x0, dtype=np.float64)
        xp[j] += h
        xm[j] -= h
        fd[j] = (f(xp) - f(xm)) / (2 * h)
    np.testing.assert_allclose(grad, fd, rtol=rtol, atol=rtol * 0.1 * np.abs(fd).max())
    return names, grad


@pytest.mark.parametrize("loss,kwargs", [("looph", {}), ("looph", {"boundary_scale": 1.7}), ("pseudo_huber", {}),
                                         ("pseudo_huber", {"boundary_scale": 0.6})])
@pytest.mark.parametrize("k,d", [(14, 5), (50, 8)])
def test_robust_losses_have_analytic_gradients_too(loss, kwargs, k, d):
    """looph and pseudo-Huber (reference: _src/optimize/loss/numpy.py:64-117), with and without a ``boundary_scale``:
    value = the objective's, gradient = its central differences (round-5 review: the analytic route refused them)."""
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from muygpys_amd.optimize import loss as L

    X, y, bi, ni = _data(50 + k, 1500, 350, k, d, np.linspace(0.8, 1.5, d))
    m = _model("matern15", True, d, np.linspace(1.0, 1.8, d), 1e-2)
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=getattr(L, loss + "_fn"), loss_kwargs=kwargs)
    _fd_check(m, obj)


@pytest.mark.parametrize("loss", ["lool", "mse", "looph"])
@pytest.mark.parametrize("scale_kind", ["analytic", "fixed"])
def test_a_free_noise_parameter_is_differentiated_as_the_reference_evaluates_it(loss, scale_kind):
    """``noise`` among the free parameters: the reference's objective gives mean and variance the TRIAL noise and computes
    the analytic sigma^2 with the model's STORED one (gp/hyperparameter/scale.py:206,214 against
    gp/noise/homoscedastic.py:112-113) -- the analytic route now follows that (two forward and two backward launches
    per evaluation) instead of refusing: value and gradient against the objective itself."""
    from muygpys_amd.gp.hyperparameter import FixedScale, Parameter
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from muygpys_amd.optimize import loss as L

    d, k = 6, 20
    X, y, bi, ni = _data(91, 1500, 300, k, d, np.linspace(0.8, 1.5, d))
    m = _model("matern25", True, d, np.linspace(1.0, 1.8, d), 1e-2, scale=FixedScale(val=0.4) if scale_kind == "fixed" else None)
    m.noise = HomoscedasticNoise(3e-2, (1e-4, 1.0))
    m._make()
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    obj = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=getattr(L, loss + "_fn"))
    names, grad = _fd_check(m, obj, rtol=5e-6)
    assert "noise" in names and abs(grad[names.index("noise")]) > 0.0


def test_analytic_gradient_refuses_what_it_does_not_differentiate():
    from muygpys_amd.gp.hyperparameter import Parameter
    from muygpys_amd.optimize import L_BFGS_B_optimize

    X, y, bi, ni = _data(9, 800, 100, 10, 4, np.ones(4))
    Xd, yd = to_dev(X, torch.float64), to_dev(y, torch.float64)
    m = _model("matern15", False, 4, 1.3, 1e-2)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd)
    with pytest.raises(ValueError, match="no target mask"):
        L_BFGS_B_optimize(m, y_b, y_nn, cross, pair, target_mask=[0], analytic_gradient=True)
    with pytest.raises(ValueError, match="loss_kwargs"):
        L_BFGS_B_optimize(m, y_b, y_nn, cross, pair, loss_kwargs={"boundary_scale": 2.0}, analytic_gradient=True)  # (lool takes none)
    cross, pair, y_b, y_nn = m.make_train_tensors(to_dev(bi), to_dev(ni), Xd, yd, materialize=True)
    m2 = _model("matern15", False, 4, 1.3, 1e-2)
    with pytest.raises(ValueError, match="lazy training tensors"):
        L_BFGS_B_optimize(m2, y_b, y_nn, cross, pair, analytic_gradient=True)
