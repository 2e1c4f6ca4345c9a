"""Optimisation chassis (reference contract: src/MuyGPyS/optimize/chassis.py:23-363).

``OptimizeFn(optimize_fn, make_obj_fn)(muygps, batch_targets, batch_nn_targets,
crosswise_diffs, pairwise_diffs, ...)`` builds the LOOCV objective from the model's closures
and hands it to the driver; it returns a NEW model with the optimised values.
"""

from __future__ import annotations

from typing import Callable, Dict, Optional

from muygpys_amd._src.optimize.chassis import _bayes_opt_optimize, _scipy_optimize

from .loss import LossFn, lool_fn
from .objective import make_loo_crossval_fn


class OptimizeFn:
    def __init__(self, optimize_fn: Callable, make_obj_fn: Callable):
        self._fn = optimize_fn
        self._make_obj_fn = make_obj_fn

    def __call__(
        self, muygps, batch_targets, batch_nn_targets, crosswise_diffs, pairwise_diffs, batch_features=None,
        loss_fn: LossFn = lool_fn, loss_kwargs: Dict = dict(), target_mask=None, verbose: bool = False, **kwargs,
    ):
        obj_fn = self.make_obj_fn(
            muygps, batch_targets, batch_nn_targets, crosswise_diffs, pairwise_diffs, batch_features=batch_features,
            target_mask=target_mask, loss_fn=loss_fn, loss_kwargs=loss_kwargs,
        )
        return self._fn(muygps, obj_fn, verbose=verbose, **kwargs)

    def make_obj_fn(
        self, muygps, batch_targets, batch_nn_targets, crosswise_diffs, pairwise_diffs, batch_features=None,
        target_mask=None, loss_fn: LossFn = lool_fn, loss_kwargs: Dict = dict(), **kwargs,
    ) -> Callable:
        """chassis.py:119-194."""
        return self._make_obj_fn(
            loss_fn, muygps.kernel.get_opt_fn(), muygps.get_opt_mean_fn(), muygps.get_opt_var_fn(),
            muygps.scale.get_opt_fn(muygps), pairwise_diffs, crosswise_diffs, batch_nn_targets, batch_targets,
            batch_features=batch_features, target_mask=target_mask, loss_kwargs=loss_kwargs,
        )


Bayes_optimize = OptimizeFn(_bayes_opt_optimize, make_loo_crossval_fn)
L_BFGS_B_optimize = OptimizeFn(_scipy_optimize, make_loo_crossval_fn)
