"""sigma^2 scale functors (reference contract: src/MuyGPyS/gp/hyperparameter/scale.py:21-219).

``ScaleFn.scale_fn`` multiplies a variance function by the stored value (captured when the
closure is built, scale.py:106); ``AnalyticScale.get_opt_fn`` returns the closed-form
optimiser ``mean_b y^T (K + eps)^-1 y / k`` used both by ``MuyGPS.optimize_scale`` and
inside the LOOCV objective.  ``DownSampleScale`` (random sub-sampling + median) is outside
the hot path and not provided.
"""

from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np


def _default_scale_backend():
    from muygpys_amd.gp.lazy_dispatch import analytic_scale_optim

    return analytic_scale_optim


class ScaleFn:
    def __init__(self, val: float = 1.0, **kwargs):
        self.val = self._check_positive_float(val)
        self._trained = False

    def _check_positive_float(self, val):
        size = getattr(val, "numel", None)
        size = size() if callable(size) else getattr(val, "size", None)
        if isinstance(val, Sequence) or (size is not None and int(size) != 1):
            raise ValueError(f"Scale parameter must be scalar, not {val}.")
        fval = float(val.reshape(-1)[0]) if size is not None else float(val)
        if fval <= 0.0:
            raise ValueError(f"Scale parameter must be positive, not {val}.")
        return fval

    def __str__(self, **kwargs):
        return f"{type(self).__name__}({self.val})"

    def _set(self, val) -> None:
        self.val = self._check_positive_float(val)
        self._trained = True

    def __call__(self):
        return self.val

    @property
    def trained(self) -> bool:
        return self._trained

    def scale_fn(self, fn: Callable) -> Callable:
        """scale.py:94-109 -- the current value is bound as the keyword default."""

        def scaled_fn(*args, scale=self(), **kwargs):
            return scale * fn(*args, **kwargs)

        return scaled_fn

    def get_opt_fn(self, muygps) -> Callable:
        def noop_scale_opt_fn(Kin, nn_targets, *args, **kwargs):
            return muygps.scale()

        return noop_scale_opt_fn


class FixedScale(ScaleFn):
    """A scale that optimisation leaves alone (scale.py:118-145)."""


class AnalyticScale(ScaleFn):
    def __init__(self, iteration_count: int = 1, _backend_fn: Optional[Callable] = None, **kwargs):
        super().__init__(**kwargs)
        self.iteration_count = iteration_count
        self._fn = _backend_fn if _backend_fn is not None else _default_scale_backend()

    def get_opt_fn(self, muygps) -> Callable:
        """scale.py:172-219.  The nugget applied here is the model's STORED noise even when the
        objective passes a trial ``noise=`` to mean/variance (reference quirk).
        The optional fixed-point iteration needs no new solves: f(s K) = f(K) / s."""

        def analytic_scale_opt_fn(Kin, nn_targets, *args, **kwargs):
            scale = self._fn(muygps.noise.perturb(Kin), nn_targets, **kwargs)
            for _ in range(1, self.iteration_count):
                scale = 0.5 * (scale + self._fn(muygps.noise.perturb(Kin), nn_targets, **kwargs) / scale)
            return scale

        return analytic_scale_opt_fn
