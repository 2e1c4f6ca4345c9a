"""The ``MuyGPS`` model object (reference contract: src/MuyGPyS/gp/muygps.py:28-567,
gp/mean.py:17-37, gp/variance.py:22-52).

It owns a kernel functor, a noise model and a sigma^2 scale and wires the closures

    mean      = backend_mean(noise.perturb(Kin), Kcross, nn_targets)
    var_opt   = backend_var(noise.perturb(Kin), Kcross, Kout)          (UNscaled: objective)
    var       = scale() * var_opt                                        (public)

``make_predict_tensors`` / ``make_train_tensors`` return (crosswise, pairwise,
[batch_targets,] batch_nn_targets) in the reference's order; under the hip backend they are
lazy handles by default so that the calls above collapse into one fused launch
(``materialize=True`` gives the reference's tensors).
"""

from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from muygpys_amd import lazy as _lazy
from muygpys_amd._src.util import auto_str
from muygpys_amd.gp import lazy_dispatch as _ld
from muygpys_amd.gp.hyperparameter import FixedScale, ScaleFn
from muygpys_amd.gp.kernels import KernelFn
from muygpys_amd.gp.noise import HomoscedasticNoise, NoiseFn


class PosteriorMean:
    """gp/mean.py:17-37."""

    def __init__(self, noise: NoiseFn, _backend_fn: Callable = _ld.posterior_mean, **kwargs):
        self._fn = noise.perturb_fn(_backend_fn)

    def __call__(self, Kin, Kcross, batch_nn_targets, **kwargs):
        return self._fn(Kin, Kcross, batch_nn_targets, **kwargs)

    def get_opt_fn(self) -> Callable:
        return self._fn


class PosteriorVariance:
    """gp/variance.py:22-52: the optimiser sees the unscaled closure, the user the scaled one."""

    def __init__(self, Kout, noise: NoiseFn, scale: ScaleFn, _backend_fn: Callable = _ld.diagonal_variance):
        perturbed = noise.perturb_fn(_backend_fn)

        def fixed_Kout_fn(Kin, Kcross, *args, **kwargs):
            return perturbed(Kin, Kcross, Kout, *args, **kwargs)

        self._opt_fn = fixed_Kout_fn
        self._fn = scale.scale_fn(fixed_Kout_fn)

    def __call__(self, Kin, Kcross, **kwargs):
        return self._fn(Kin, Kcross, **kwargs)

    def get_opt_fn(self) -> Callable:
        return self._opt_fn


@auto_str
class MuyGPS:
    def __init__(
        self,
        kernel: KernelFn,
        noise: NoiseFn = HomoscedasticNoise(0.0, "fixed"),
        scale: ScaleFn = FixedScale(),
        _backend_mean_fn: Callable = _ld.posterior_mean,
        _backend_var_fn: Callable = _ld.diagonal_variance,
        _backend_fast_mean_fn: Callable = _ld.fast_posterior_mean,
        _backend_fast_precompute_fn: Callable = _ld.fast_posterior_mean_precompute,
    ):
        self.kernel = kernel
        self.scale = scale
        self.noise = noise
        self._backend_mean_fn = _backend_mean_fn
        self._backend_var_fn = _backend_var_fn
        self._backend_fast_mean_fn = _backend_fast_mean_fn
        self._backend_fast_precompute_fn = _backend_fast_precompute_fn
        self._make()

    def _make(self) -> None:
        self.kernel._make()
        self._mean_fn = PosteriorMean(self.noise, _backend_fn=self._backend_mean_fn)
        self._var_fn = PosteriorVariance(self.kernel.Kout(), self.noise, self.scale, _backend_fn=self._backend_var_fn)
        self._fast_precompute_fn = self.noise.perturb_fn(self._backend_fast_precompute_fn)
        self._fast_posterior_mean_fn = self._backend_fast_mean_fn

    def set_params(self, **kwargs) -> None:
        self.kernel.set_params(**kwargs)
        self._make()

    def fixed(self) -> bool:
        """muygps.py:128-143."""
        if any(not p.fixed() for p in self.kernel._hyperparameters.values()):
            return False
        return self.noise.fixed()

    def get_opt_params(self) -> Tuple[List[str], np.ndarray, np.ndarray]:
        """muygps.py:145-162."""
        names, params, bounds = self.kernel.get_opt_params()
        self.noise.append_lists(names, params, bounds)
        return names, np.array(params, dtype=np.float64), np.array(bounds, dtype=np.float64)

    def posterior_mean(self, Kin, Kcross, batch_nn_targets):
        """muygps.py:164-211."""
        return self._mean_fn(Kin, Kcross, batch_nn_targets)

    def posterior_variance(self, Kin, Kcross):
        """muygps.py:213-259 -- already multiplied by sigma^2."""
        return self._var_fn(Kin, Kcross)

    def fast_coefficients(self, Kin, train_nn_targets_fast):
        """muygps.py:261-298."""
        return self._fast_precompute_fn(Kin, train_nn_targets_fast)

    def fast_posterior_mean(self, Kcross, coeffs_tensor):
        """muygps.py:300-341."""
        return self._fast_posterior_mean_fn(Kcross, coeffs_tensor)

    def get_opt_mean_fn(self) -> Callable:
        return self._mean_fn.get_opt_fn()

    def get_opt_var_fn(self) -> Callable:
        return self._var_fn.get_opt_fn()

    def optimize_scale(self, pairwise_diffs, nn_targets):
        """muygps.py:373-403: sigma^2 <- scale.get_opt_fn(self)(kernel(pairwise), nn_targets)."""
        Kin = self.kernel(pairwise_diffs)
        opt_fn = self.scale.get_opt_fn(self)
        self.scale._set(opt_fn(Kin, nn_targets))
        self._make()
        return self

    def make_predict_tensors(
        self, batch_indices, batch_nn_indices, test_features, train_features, train_targets, materialize: bool = False,
    ):
        """muygps.py:406-475 -> (crosswise, pairwise, batch_nn_targets)."""
        if test_features is None:
            test_features = train_features
        lazy = not materialize
        crosswise = self.kernel.deformation.crosswise_tensor(
            test_features, train_features, batch_indices, batch_nn_indices, lazy=lazy
        )
        pairwise = self.kernel.deformation.pairwise_tensor(train_features, batch_nn_indices, lazy=lazy)
        nn_targets = (
            _lazy.LazyTargets(train_targets, batch_nn_indices) if lazy else train_targets[batch_nn_indices]
        )
        return crosswise, pairwise, nn_targets

    def make_train_tensors(
        self, batch_indices, batch_nn_indices, train_features, train_targets, materialize: bool = False, **kwargs
    ):
        """muygps.py:478-551 -> (crosswise, pairwise, batch_targets, batch_nn_targets)."""
        crosswise, pairwise, nn_targets = self.make_predict_tensors(
            batch_indices, batch_nn_indices, train_features, train_features, train_targets, materialize=materialize
        )
        return crosswise, pairwise, train_targets[batch_indices], nn_targets

    def __eq__(self, rhs) -> bool:
        if not isinstance(rhs, self.__class__):
            return False
        mine, theirs = self.kernel._hyperparameters, rhs.kernel._hyperparameters
        return (
            all(np.all(mine[h]() == theirs[h]()) for h in mine)
            and np.all(self.noise() == rhs.noise())
            and self.scale() == rhs.scale()
        )
