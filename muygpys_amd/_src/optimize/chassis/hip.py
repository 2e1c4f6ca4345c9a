"""Outer optimisation drivers (reference: src/MuyGPyS/_src/optimize/chassis/numpy.py:15-149).

These are host-side loops around the (GPU) objective, one scalar per evaluation:

* ``_scipy_optimize``: scipy's L-BFGS-B with finite-difference gradients, exactly the
  reference's call (numpy.py:57-81).
* ``_bayes_opt_optimize``: the reference delegates to the third-party package
  ``bayesian-optimization`` (>= 1.4.2, pyproject.toml:43), which is not vendored and not
  installed in the build image.  What follows is a self-contained restatement of that
  package's published default algorithm -- probe x0, ``init_points`` uniform samples, then
  ``n_iter`` rounds of: fit a GP (Matern nu=2.5, alpha=1e-6, normalised targets) to the
  evaluations and maximise the UCB acquisition (kappa = 2.576) by 10 000 random candidates
  refined with L-BFGS-B from the best ten.  The surrogate is written out (``_SurrogateGP``:
  warm-started length-scale fit, ``K^-1`` reused, candidates drawn / scored / ranked on the
  device, the ten best polished in one stacked L-BFGS-B run) so that a suggestion costs a
  couple of milliseconds, not the ~200 ms of scikit-learn's regressor, next to a 3 ms objective.  PARITY UNPINNED: no reference
  test pins a Bayes-opt trajectory or value; only behaviour (recovers
  planted hyper-parameters within the reference's statistical tolerances) is checked.
"""

from __future__ import annotations

from copy import deepcopy
from typing import Callable, Dict, List, Tuple

import numpy as np


def _as_float(x) -> float:
    return float(x.item()) if hasattr(x, "item") else float(x)


def _new_muygps(muygps, x0_names, bounds, opt_dict):
    """numpy.py:15-34: clamp to the bounds, write into a deep copy, rebuild its closures."""
    ret = deepcopy(muygps)
    for i, key in enumerate(x0_names):
        lb, ub = bounds[i]
        val = min(max(float(opt_dict[key]), lb), ub)
        if key == "noise":
            ret.noise._set_val(val)
        else:
            ret.kernel._hyperparameters[key]._set_val(val)
    ret._make()
    return ret


def _obj_fn_adapter(obj_fn, x0_names):
    """numpy.py:37-42: keyword objective (maximised) -> array objective (minimised)."""

    def array_obj_fn(x_array, *args):
        return -_as_float(obj_fn(*args, **{h: float(x_array[i]) for i, h in enumerate(x0_names)}))

    return array_obj_fn


def _get_opt_lists(muygps, verbose: bool = False):
    x0_names, x0, bounds = muygps.get_opt_params()
    # Under sharded reductions every rank must walk the SAME trajectory (equal numbers of all-reduces, every
    # shard evaluated at the same hyper-parameters): the start point is rank 0's, whenever and however this
    # rank's model was built ("sample" / "log_sample" values drawn before the block was entered differ per rank).
    from muygpys_amd import distributed as _D

    if _D.reductions_active() and len(x0):
        x0 = _D.broadcast_vector(np.asarray(x0, dtype=np.float64), _D.active_group())
    if verbose:
        print(f"parameters to be optimized: {x0_names}")
        print(f"bounds: {bounds}")
        print(f"initial x0: {x0}")
    return x0_names, np.asarray(x0, dtype=np.float64), np.asarray(bounds, dtype=np.float64).reshape(-1, 2)


def _analytic_value_and_grad(muygps, obj_fn, x0_names):
    """``x -> (loss, d loss / d x)`` for scipy's ``jac=True``: one fused LOOCV evaluation and one backward launch
    (``muygpys_amd.fused.loocv_value_and_grad``) instead of ``len(x) + 1`` evaluations per iteration.  The objective
    must be what ``make_loo_crossval_fn`` builds on lazy handles (``obj_fn.loocv_context``); the loss any of
    ``lool_fn``, ``mse_fn``, ``pseudo_huber_fn``, ``looph_fn`` (with their ``boundary_scale``); the free parameters
    length scales (``length_scale`` / ``length_scale{i}``) of a closed-form kernel and / or the homoscedastic ``noise``
    -- anything else raises (there is no silent fall-back to finite differences: the caller asked for an analytic
    gradient).  What the reference's torch autograd route differentiates (torch/muygps_layer.py:129-164), here for the
    numpy-style chassis (_src/optimize/chassis/numpy.py:57-81).

    A free ``noise`` follows the reference's objective to the letter: mean and variance take the TRIAL value, the
    analytic sigma^2 is computed with the model's STORED noise (gp/hyperparameter/scale.py:206,214 against
    gp/noise/homoscedastic.py:112-113) -- two forward and two backward launches per evaluation then, and the noise
    gradient is the trial value's alone.

    The sigma^2 inside ``lool`` / ``looph`` is whatever the objective was built with, and the gradient follows it: the
    closed form of ``AnalyticScale`` (any ``iteration_count``: the fixed-point passes are scalar algebra on the first
    value and are differentiated as such) or the constant of ``FixedScale`` / a plain ``ScaleFn`` (no cotangent through
    ``y^T K^-1 y``).  A ``scale_fn`` that is neither of the two closures ``ScaleFn.get_opt_fn`` returns is refused."""
    from muygpys_amd import distributed as D
    from muygpys_amd import lazy, lazy_eval
    from muygpys_amd.fused import loocv_value_and_grad
    from muygpys_amd.optimize.loss import lool_fn, looph_fn, mse_fn, pseudo_huber_fn

    ctx = getattr(obj_fn, "loocv_context", None)
    if ctx is None:
        raise ValueError("analytic_gradient=True: the objective was not built by make_loo_crossval_fn")
    loss = {id(lool_fn): "lool", id(mse_fn): "mse", id(pseudo_huber_fn): "pseudo_huber", id(looph_fn): "looph"}.get(id(ctx["loss_fn"]))
    if loss is None or ctx["target_mask"] is not None:
        raise ValueError("analytic_gradient=True: the gradient is written out for lool_fn, mse_fn, pseudo_huber_fn and "
                         "looph_fn (no target mask)")
    unknown = set(ctx.get("loss_kwargs") or {}) - ({"boundary_scale"} if loss in ("pseudo_huber", "looph") else set())
    if unknown:
        raise ValueError(f"analytic_gradient=True: loss_kwargs {sorted(unknown)} are not differentiated")
    boundary_scale = (ctx.get("loss_kwargs") or {}).get("boundary_scale")
    scale_mode = _scale_mode(muygps, ctx.get("scale_fn")) if loss in ("lool", "looph") else ("analytic", 1)
    pair, cross, nn_t = ctx["pairwise_diffs"], ctx["crosswise_diffs"], ctx["batch_nn_targets"]
    if not (isinstance(pair, lazy.LazyDiffs) and isinstance(cross, lazy.LazyDiffs) and isinstance(nn_t, lazy.LazyTargets)):
        raise ValueError("analytic_gradient=True: needs the lazy training tensors (MuyGPS.make_train_tensors under "
                         "config.state.lazy_tensors / integration.install())")
    if cross.data is not pair.nn_data and cross.data.data_ptr() != pair.nn_data.data_ptr():
        raise ValueError("analytic_gradient=True: LOOCV differentiates one table (query rows = training rows)")
    index, noise_at = {}, None
    for j, name in enumerate(x0_names):
        if name == "length_scale":
            index[j] = 0
        elif name.startswith("length_scale") and name[len("length_scale"):].isdigit():
            index[j] = int(name[len("length_scale"):])
        elif name == "noise":
            noise_at = j
        else:
            raise ValueError(f"analytic_gradient=True: {name!r} is neither a length scale nor the homoscedastic noise")
    stored = muygps.noise()
    if getattr(stored, "ndim", 0) >= 1 and getattr(stored, "numel", lambda: 1)() > 1:
        raise ValueError("analytic_gradient=True: homoscedastic noise")
    stored = float(stored.item()) if hasattr(stored, "item") else float(stored)
    reduce_fn = D.reduce_if_sharded_ if D.reductions_active() else None

    def value_and_grad(x_array, *args):
        hyper = {h: float(x_array[i]) for i, h in enumerate(x0_names) if i != noise_at}
        Kin = ctx["kernel_fn"](pair, **hyper)
        if not isinstance(Kin, lazy.LazyCov):
            raise ValueError("analytic_gradient=True: the kernel did not stay a lazy handle")
        spec = lazy_eval._spec(Kin)
        spec.noise = stored if noise_at is None else float(x_array[noise_at])
        value, g_ls, g_noise = loocv_value_and_grad(spec, pair.nn_data, nn_t.targets, cross.data_indices, pair.nn_indices,
                                                    loss=loss, reduce_fn=reduce_fn, scale=scale_mode,
                                                    boundary_scale=boundary_scale,
                                                    sigma_noise=stored if noise_at is not None else None)
        grad = [g_noise if j == noise_at else g_ls[index[j]] for j in range(len(x0_names))]
        return value, np.array(grad, dtype=np.float64)

    return value_and_grad


def _scale_mode(muygps, scale_fn):
    """Which sigma^2 the objective's ``lool`` divides by: ``("analytic", iteration_count)`` or ``("fixed", value)``.
    Read from the closure the objective was BUILT with (gp/hyperparameter/scale.py:60-63,172-219 of the reference:
    ``noop_scale_opt_fn`` returns ``muygps.scale()``, ``analytic_scale_opt_fn`` the closed form), cross-checked
    against the model's scale object; anything else cannot be differentiated here and raises."""
    from muygpys_amd.gp.hyperparameter.scale import AnalyticScale

    name = getattr(scale_fn, "__name__", None)
    scale = getattr(muygps, "scale", None)
    if name == "analytic_scale_opt_fn" and isinstance(scale, AnalyticScale):
        if np.asarray(scale.val).size != 1:
            raise ValueError("analytic_gradient=True: the analytic scale of a vector-valued sigma^2 is not written out")
        return ("analytic", max(int(scale.iteration_count), 1))
    if name == "noop_scale_opt_fn" and scale is not None and not isinstance(scale, AnalyticScale):
        return ("fixed", _as_float(scale()))
    raise ValueError(
        "analytic_gradient=True: lool_fn's gradient is written out for the sigma^2 of AnalyticScale.get_opt_fn "
        f"or a fixed ScaleFn value; the objective was built with scale_fn={name!r} on a {type(scale).__name__} model"
    )


def _scipy_optimize(muygps, obj_fn: Callable, verbose: bool = False, analytic_gradient: bool = False, **kwargs):
    """numpy.py:57-81.  ``analytic_gradient`` (round 5, opt-in; default: the reference's finite differences, the same
    trajectory as the reference's): hand scipy the analytic gradient of the LOOCV loss."""
    from scipy import optimize as opt

    x0_names, x0, bounds = _get_opt_lists(muygps, verbose=verbose)
    if analytic_gradient:
        optres = opt.minimize(_analytic_value_and_grad(muygps, obj_fn, x0_names), x0, jac=True, method="L-BFGS-B",
                              bounds=bounds, **kwargs)
    else:
        optres = opt.minimize(_obj_fn_adapter(obj_fn, x0_names), x0, method="L-BFGS-B", bounds=bounds, **kwargs)
    if verbose:
        print(f"optimizer results: \n{optres}")
    return _new_muygps(muygps, x0_names, bounds, {n: optres.x[i] for i, n in enumerate(x0_names)})


class _SurrogateGP:
    """The surrogate of the Bayes driver: a zero-mean GP with a Matern-5/2 kernel of one length scale on
    the evaluated points, nugget 1e-6, targets standardised (the published defaults of the
    ``bayesian-optimization`` package's scikit-learn regressor).  Written out because the driver is
    host-bound otherwise: scikit-learn's regressor with five restarts and a per-point ``predict`` made a
    suggestion cost ~200 ms against a 3 ms objective.  Here the length scale is the one free parameter
    (log-marginal likelihood maximised by L-BFGS-B with its analytic derivative, WARM-STARTED from the previous
    optimum; a second, random start on the first fit and every eighth after it), ``K^-1`` is kept, and mean /
    deviation / their gradients are closed forms evaluated for all candidates at once -- the thousands of random
    candidates on the device (:meth:`top_candidates`), the handful being polished on the host (:meth:`ucb`)."""

    ALPHA = 1e-6
    S5 = 5.0 ** 0.5

    def __init__(self, rng):
        self.rng = rng
        self.log_ell = 0.0
        self.fits = 0

    @classmethod
    def _kernel(cls, r, ell):
        t = cls.S5 * r / ell
        return (1.0 + t + t * t / 3.0) * np.exp(-t)

    def _neg_lml(self, log_ell, R, y):
        from scipy.linalg import cho_factor, cho_solve

        ell = float(np.exp(np.asarray(log_ell).reshape(-1)[0]))
        t = self.S5 * R / ell
        e = np.exp(-t)
        K = (1.0 + t + t * t / 3.0) * e
        K[np.diag_indices_from(K)] += self.ALPHA
        try:
            c = cho_factor(K, lower=True)
        except np.linalg.LinAlgError:
            return 1e25, np.zeros(1)
        a = cho_solve(c, y)
        lml = -0.5 * float(y @ a) - float(np.log(np.diag(c[0])).sum()) - 0.5 * len(y) * np.log(2 * np.pi)
        dK = (t * t / 3.0) * (1.0 + t) * e  # dK / d log(ell)
        g = 0.5 * float(a @ dK @ a) - 0.5 * float(np.trace(cho_solve(c, dK)))
        return -lml, np.array([-g])

    def fit(self, X, y):
        from scipy.linalg import cho_factor, cho_solve
        from scipy.optimize import minimize

        self.X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        self.mu, self.sd = float(y.mean()), float(y.std()) or 1.0
        yn = (y - self.mu) / self.sd
        R = np.sqrt(np.maximum(((self.X[:, None, :] - self.X[None, :, :]) ** 2).sum(-1), 0.0))
        starts = [self.log_ell]
        if self.fits % 8 == 0:
            starts.append(float(self.rng.uniform(np.log(1e-2), np.log(1e2))))
        self.fits += 1
        best = None
        for start in starts:
            res = minimize(self._neg_lml, [start], args=(R, yn), jac=True, method="L-BFGS-B",
                           bounds=[(np.log(1e-5), np.log(1e5))])
            if best is None or res.fun < best.fun:
                best = res
        self.log_ell = float(best.x[0])
        self.ell = float(np.exp(self.log_ell))
        K = self._kernel(R, self.ell)
        K[np.diag_indices_from(K)] += self.ALPHA
        self.chol = cho_factor(K, lower=True)
        self.alpha = cho_solve(self.chol, yn)
        self.Kinv = cho_solve(self.chol, np.eye(len(yn)))
        return self

    def ucb(self, Xc, kappa, want_grad=False):
        """UCB (and its gradient) of the candidates Xc (m, p), in the units of the objective."""
        Xc = np.atleast_2d(Xc)
        if not want_grad:
            return self._ucb_many(Xc, kappa)
        diff = Xc[:, None, :] - self.X[None, :, :]                      # (m, n, p): a handful of points
        t = np.sqrt(np.einsum("mnp,mnp->mn", diff, diff)) * (self.S5 / self.ell)
        e = np.exp(-t)
        k = (1.0 + t + t * t / 3.0) * e
        v = k @ self.Kinv                                               # (m, n) = (K^-1 k)^T
        var = np.maximum(1.0 - np.einsum("mn,mn->m", k, v), 1e-18)
        std = np.sqrt(var)
        val = self.mu + self.sd * (k @ self.alpha + kappa * std)
        # dk/dx = -(5 / (3 ell^2)) (1 + t) e^{-t} (x - x_i);  d mean = dk . alpha,  d var = -2 dk . K^-1 k
        w = (-(5.0 / (3.0 * self.ell**2)) * self.sd) * (1.0 + t) * e * (self.alpha[None, :] - (kappa / std)[:, None] * v)
        grad = np.einsum("mn,mnp->mp", w, diff)
        return val, grad

    def _device(self):
        import torch

        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def _ucb_torch(self, C, kappa):
        """UCB of the rows of the device tensor C in one pass of tensor operations: pairwise distances, kernel, one
        product with K^-1 (n <= a few dozen evaluated points).  fp64 throughout."""
        import torch

        to = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=C.device)  # noqa: E731
        t = torch.cdist(C, to(self.X)) * (self.S5 / self.ell)
        k = (1.0 + t + t * t / 3.0) * torch.exp(-t)
        var = (1.0 - ((k @ to(self.Kinv)) * k).sum(1)).clamp_min(1e-18)
        return self.mu + self.sd * (k @ to(self.alpha) + kappa * var.sqrt())

    def _ucb_many(self, Xc, kappa):
        import torch

        C = torch.as_tensor(np.ascontiguousarray(Xc), dtype=torch.float64, device=self._device())
        return self._ucb_torch(C, kappa).cpu().numpy()

    def top_candidates(self, count: int, keep: int, bounds: np.ndarray, kappa: float, generator):
        """``count`` uniform candidates in the box, drawn, scored and ranked ON THE DEVICE; only the best ``keep`` come
        back: (points (keep, p), values (keep,)), best first.  One host <- device transfer of keep (p + 1) doubles."""
        import torch

        dev = self._device()
        lo = torch.as_tensor(bounds[:, 0], dtype=torch.float64, device=dev)
        hi = torch.as_tensor(bounds[:, 1], dtype=torch.float64, device=dev)
        C = lo + (hi - lo) * torch.rand((count, bounds.shape[0]), dtype=torch.float64, device=dev, generator=generator)
        vals, idx = torch.topk(self._ucb_torch(C, kappa), min(keep, count))
        out = torch.cat([C[idx], vals[:, None]], dim=1).cpu().numpy()
        return out[:, :-1], out[:, -1]


class _UCBBayesOpt:
    """Minimal GP-UCB maximiser over a box (see the module docstring for provenance).

    One acquisition step (round 6; 13.9 ms per trial before, on the host of the GPU box): warm-started surrogate fit,
    10 000 candidates drawn / scored / ranked on the device, the ten best polished TOGETHER -- the sum of their
    negative UCBs is separable, so ONE bounded L-BFGS-B run over the stacked 10 p coordinates refines all ten with a
    tenth of the function evaluations ten separate runs take (each evaluation is one vectorised closed form).  Under
    sharded reductions rank 0 alone runs it and broadcasts the point (``distributed.broadcast_vector``): the other
    ranks neither repeat the work nor can a last-bit difference between devices make them probe another point."""

    CANDIDATES, POLISHED, POLISH_EVALS = 10000, 10, 30

    def __init__(self, f: Callable, names: List[str], bounds: np.ndarray, random_state=None, verbose: int = 0):
        self.f, self.names, self.bounds = f, names, bounds
        self.rng = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) else random_state
        self.verbose = verbose
        self.X: List[np.ndarray] = []
        self.y: List[float] = []
        self.gp = _SurrogateGP(self.rng)
        self.suggest_seconds = 0.0
        self._gen = None

    def probe(self, x: np.ndarray) -> float:
        val = _as_float(self.f(**{n: float(x[i]) for i, n in enumerate(self.names)}))
        self.X.append(np.asarray(x, dtype=np.float64))
        self.y.append(val)
        if self.verbose:
            print(f"| {len(self.y):4d} | {val: .6g} | " + " | ".join(f"{v:.6g}" for v in x))
        return val

    def _sample(self, n: int) -> np.ndarray:
        return self.rng.uniform(self.bounds[:, 0], self.bounds[:, 1], size=(n, len(self.names)))

    def _generator(self):
        """The device-side stream of candidate draws, seeded once from the driver's RandomState."""
        import torch

        if self._gen is None:
            self._gen = torch.Generator(device=self.gp._device())
            self._gen.manual_seed(int(self.rng.randint(0, 2**31 - 1)))
        return self._gen

    def _suggest(self, kappa: float) -> np.ndarray:
        import time

        from scipy.optimize import minimize

        from muygpys_amd import distributed as _D

        sharded = _D.reductions_active()
        group = _D.active_group() if sharded else None
        if sharded and _D._world(group)[0] != 0:
            return _D.broadcast_vector(np.zeros(len(self.names)), group)
        t0 = time.perf_counter()
        self.gp.fit(np.array(self.X), np.array(self.y))
        seeds, seed_vals = self.gp.top_candidates(self.CANDIDATES, self.POLISHED, self.bounds, kappa, self._generator())
        m, p = seeds.shape

        def neg(z):
            v, g = self.gp.ucb(z.reshape(m, p), kappa, want_grad=True)
            return -float(v.sum()), -g.reshape(-1)

        res = minimize(neg, seeds.reshape(-1), jac=True, bounds=np.tile(self.bounds, (m, 1)), method="L-BFGS-B",
                       options={"maxfun": self.POLISH_EVALS})
        polished = np.clip(res.x.reshape(m, p), self.bounds[:, 0], self.bounds[:, 1])
        vals = self.gp.ucb(polished, kappa, want_grad=True)[0]
        best = int(np.argmax(vals))
        best_x = polished[best] if vals[best] > seed_vals[0] else seeds[0]
        self.suggest_seconds += time.perf_counter() - t0
        best_x = np.clip(best_x, self.bounds[:, 0], self.bounds[:, 1])
        return _D.broadcast_vector(best_x, group) if sharded else best_x

    def maximize(self, init_points: int = 5, n_iter: int = 20, kappa: float = 2.576, **ignored) -> Dict:
        for x in self._sample(init_points):
            self.probe(x)
        for _ in range(n_iter):
            self.probe(self._suggest(kappa))
        best = int(np.argmax(self.y))
        return {"target": self.y[best], "params": {n: float(self.X[best][i]) for i, n in enumerate(self.names)}}


def _bayes_opt_optimize(muygps, obj_fn: Callable, verbose: bool = False, **kwargs):
    """numpy.py:119-149: probe x0 first, then init_points (default 5) + n_iter (default 20).

    ``log_bounds=True`` (round 5, opt-in; not in the reference): search in the logarithms of the parameters -- for
    scale-like parameters whose bounds span decades (length scales in (0.1, 10)) the uniform initial points and the
    stationary surrogate then see every decade alike, where the linear box puts nine samples in ten above 1.  With the
    reference's defaults on BASELINE config 4 (eight length scales, 25 trials) the linear search never improved on its
    start point; see docs/HISTORY.md sec. 4.5."""
    x0_names, x0, bounds = _get_opt_lists(muygps, verbose=verbose)
    if kwargs.get("random_state") is None:
        from muygpys_amd import distributed as _D

        if _D.reductions_active():  # every rank must propose the same points: rank 0's seed
            kwargs["random_state"] = _D.synchronized_seed(_D._ACTIVE["group"])
    log_bounds = bool(kwargs.get("log_bounds", False))
    f, box, start = obj_fn, bounds, x0
    if log_bounds:
        if not (np.all(bounds > 0) and np.all(np.asarray(x0) > 0)):
            raise ValueError("log_bounds=True needs positive bounds and a positive start point")
        box, start = np.log(bounds), np.log(x0)

        def f(**z):  # noqa: F811  (the objective in the logarithms of its parameters)
            return obj_fn(**{name: float(np.exp(v)) for name, v in z.items()})

    optimizer = _UCBBayesOpt(
        f, x0_names, box, random_state=kwargs.get("random_state"),
        verbose=kwargs.get("verbose", 2 if verbose else 0) if not isinstance(kwargs.get("verbose"), bool) else int(verbose),
    )
    optimizer.probe(start)
    maximize_kwargs = {k: kwargs[k] for k in ("init_points", "n_iter", "kappa") if k in kwargs}
    maximize_kwargs.setdefault("init_points", 5)
    maximize_kwargs.setdefault("n_iter", 20)
    best = optimizer.maximize(**maximize_kwargs)
    params = {n: (float(np.exp(v)) if log_bounds else v) for n, v in best["params"].items()}
    return _new_muygps(muygps, x0_names, bounds, params)
