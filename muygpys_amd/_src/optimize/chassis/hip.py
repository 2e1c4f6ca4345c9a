"""Outer optimisation drivers (reference: src/MuyGPyS/_src/optimize/chassis/numpy.py:15-149).

These are host-side loops around the (GPU) objective, one scalar per evaluation:

* ``_scipy_optimize``: scipy's L-BFGS-B with finite-difference gradients, exactly the
  reference's call (numpy.py:57-81).
* ``_bayes_opt_optimize``: the reference delegates to the third-party package
  ``bayesian-optimization`` (>= 1.4.2, pyproject.toml:43), which is not vendored and not
  installed in the build image.  What follows is a self-contained restatement of that
  package's published default algorithm -- probe x0, ``init_points`` uniform samples, then
  ``n_iter`` rounds of: fit a GP (Matern nu=2.5, alpha=1e-6, normalised targets, 5 optimiser
  restarts) to the evaluations and maximise the UCB acquisition (kappa = 2.576) by 10 000
  random candidates refined with L-BFGS-B from the best ten.  PARITY UNPINNED: no reference
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
    if verbose:
        print(f"parameters to be optimized: {x0_names}")
        print(f"bounds: {bounds}")
        print(f"initial x0: {x0}")
    return x0_names, np.asarray(x0, dtype=np.float64), np.asarray(bounds, dtype=np.float64).reshape(-1, 2)


def _scipy_optimize(muygps, obj_fn: Callable, verbose: bool = False, **kwargs):
    from scipy import optimize as opt

    x0_names, x0, bounds = _get_opt_lists(muygps, verbose=verbose)
    optres = opt.minimize(_obj_fn_adapter(obj_fn, x0_names), x0, method="L-BFGS-B", bounds=bounds, **kwargs)
    if verbose:
        print(f"optimizer results: \n{optres}")
    return _new_muygps(muygps, x0_names, bounds, {n: optres.x[i] for i, n in enumerate(x0_names)})


class _UCBBayesOpt:
    """Minimal GP-UCB maximiser over a box (see the module docstring for provenance)."""

    def __init__(self, f: Callable, names: List[str], bounds: np.ndarray, random_state=None, verbose: int = 0):
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import Matern

        self.f, self.names, self.bounds = f, names, bounds
        self.rng = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) else random_state
        self.verbose = verbose
        self.X: List[np.ndarray] = []
        self.y: List[float] = []
        self.gp = GaussianProcessRegressor(
            kernel=Matern(nu=2.5), alpha=1e-6, normalize_y=True, n_restarts_optimizer=5, random_state=self.rng
        )

    def probe(self, x: np.ndarray) -> float:
        val = _as_float(self.f(**{n: float(x[i]) for i, n in enumerate(self.names)}))
        self.X.append(np.asarray(x, dtype=np.float64))
        self.y.append(val)
        if self.verbose:
            print(f"| {len(self.y):4d} | {val: .6g} | " + " | ".join(f"{v:.6g}" for v in x))
        return val

    def _sample(self, n: int) -> np.ndarray:
        return self.rng.uniform(self.bounds[:, 0], self.bounds[:, 1], size=(n, len(self.names)))

    def _suggest(self, kappa: float) -> np.ndarray:
        import warnings

        from scipy.optimize import minimize

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.gp.fit(np.array(self.X), np.array(self.y))

        def ucb(x):
            mean, std = self.gp.predict(np.atleast_2d(x), return_std=True)
            return mean + kappa * std

        cand = self._sample(10000)
        vals = ucb(cand)
        best_x, best_v = cand[int(np.argmax(vals))], float(np.max(vals))
        for seed in cand[np.argsort(vals)[-10:]]:
            res = minimize(lambda x: -float(ucb(x)[0]), seed, bounds=self.bounds, method="L-BFGS-B")
            if res.success and -res.fun > best_v:
                best_x, best_v = res.x, -res.fun
        return np.clip(best_x, self.bounds[:, 0], self.bounds[:, 1])

    def maximize(self, init_points: int = 5, n_iter: int = 20, kappa: float = 2.576, **ignored) -> Dict:
        for x in self._sample(init_points):
            self.probe(x)
        for _ in range(n_iter):
            self.probe(self._suggest(kappa))
        best = int(np.argmax(self.y))
        return {"target": self.y[best], "params": {n: float(self.X[best][i]) for i, n in enumerate(self.names)}}


def _bayes_opt_optimize(muygps, obj_fn: Callable, verbose: bool = False, **kwargs):
    """numpy.py:119-149: probe x0 first, then init_points (default 5) + n_iter (default 20)."""
    x0_names, x0, bounds = _get_opt_lists(muygps, verbose=verbose)
    if kwargs.get("random_state") is None:
        from muygpys_amd import distributed as _D

        if _D.reductions_active():  # every rank must propose the same points: rank 0's seed
            kwargs["random_state"] = _D.synchronized_seed(_D._ACTIVE["group"])
    optimizer = _UCBBayesOpt(
        obj_fn, x0_names, bounds, random_state=kwargs.get("random_state"),
        verbose=kwargs.get("verbose", 2 if verbose else 0) if not isinstance(kwargs.get("verbose"), bool) else int(verbose),
    )
    optimizer.probe(x0)
    maximize_kwargs = {k: kwargs[k] for k in ("init_points", "n_iter", "kappa") if k in kwargs}
    maximize_kwargs.setdefault("init_points", 5)
    maximize_kwargs.setdefault("n_iter", 20)
    best = optimizer.maximize(**maximize_kwargs)
    return _new_muygps(muygps, x0_names, bounds, best["params"])
