"""The leave-one-out cross-validation objective (reference contract:
src/MuyGPyS/optimize/objective.py:20-118)."""

from __future__ import annotations

from typing import Callable, Dict, Optional

from .loss import LossFn


def make_kernels_fn(kernel_fn: Callable, pairwise_diffs, crosswise_diffs) -> Callable:
    """objective.py:108-118."""

    def kernels_fn(*args, **kwargs):
        Kin = kernel_fn(pairwise_diffs, *args, **kwargs)
        Kcross = kernel_fn(crosswise_diffs, *args, **kwargs)
        return Kin, Kcross

    return kernels_fn


def make_loo_crossval_fn(
    loss_fn: LossFn, kernel_fn: Callable, mean_fn: Callable, var_fn: Callable, scale_fn: Callable, pairwise_diffs,
    crosswise_diffs, batch_nn_targets, batch_targets, batch_features=None, target_mask=None,
    loss_kwargs: Dict = dict(),
) -> Callable:
    """objective.py:20-105: obj(**hyper) = -loss(mean, targets, var, sigma^2) with the kernel
    tensors re-evaluated at the trial hyper-parameters."""
    kernels_fn = make_kernels_fn(kernel_fn, pairwise_diffs, crosswise_diffs)
    predict_and_loss_fn = loss_fn.make_predict_and_loss_fn(
        mean_fn, var_fn, scale_fn, batch_nn_targets, batch_targets, target_mask=target_mask, **loss_kwargs
    )

    def obj_fn(*args, **kwargs):
        Kin, Kcross = kernels_fn(*args, batch_features=batch_features, **kwargs)
        return predict_and_loss_fn(Kin, Kcross, *args, **kwargs)

    # what the objective was built from, for a driver that differentiates it analytically instead of by finite
    # differences (_src/optimize/chassis/hip.py: _scipy_optimize(..., analytic_gradient=True))
    obj_fn.loocv_context = dict(loss_fn=loss_fn, kernel_fn=kernel_fn, scale_fn=scale_fn, pairwise_diffs=pairwise_diffs,
                                crosswise_diffs=crosswise_diffs, batch_nn_targets=batch_nn_targets,
                                batch_targets=batch_targets, target_mask=target_mask, loss_kwargs=dict(loss_kwargs))
    return obj_fn
