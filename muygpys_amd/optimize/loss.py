"""Loss functors and predict-and-loss closures (reference contract:
src/MuyGPyS/optimize/loss.py:26-396).

``LossFn(loss_fn, make_predict_and_loss_fn)`` pairs a backend loss with the recipe that
evaluates it inside the objective: *raw* losses (mse, pseudo-Huber, cross-entropy) need only
the posterior mean; *variance* losses (lool, looph) also need the unscaled posterior variance
and the analytic sigma^2.  Every closure returns MINUS the loss (loss.py:94,174) because the
Bayesian optimiser maximises.
"""

from __future__ import annotations

from typing import Callable, Optional

from muygpys_amd._src.optimize.loss import (
    _cross_entropy_fn,
    _looph_fn,
    _lool_fn,
    _lool_fn_unscaled,
    _mse_fn,
    _pseudo_huber_fn,
)


def make_raw_predict_and_loss_fn(
    loss_fn: Callable, mean_fn: Callable, var_fn: Callable, scale_fn: Callable, batch_nn_targets, batch_targets,
    target_mask=None, **loss_kwargs,
) -> Callable:
    """loss.py:26-96."""

    def predict_and_loss_fn(Kin, Kcross, *args, **kwargs):
        predictions = mean_fn(Kin, Kcross, batch_nn_targets, **kwargs)
        if target_mask is not None:
            predictions = predictions[:, target_mask]
        return -loss_fn(predictions, batch_targets, **loss_kwargs)

    return predict_and_loss_fn


def make_var_predict_and_loss_fn(
    loss_fn: Callable, mean_fn: Callable, var_fn: Callable, scale_fn: Callable, batch_nn_targets, batch_targets,
    target_mask=None, **loss_kwargs,
) -> Callable:
    """loss.py:99-178.  Evaluation order as in the reference: mean, scale, variance -- under
    the hip backend the three share one fused launch when the tensors are lazy handles."""

    def predict_and_loss_fn(Kin, Kcross, *args, **kwargs):
        predictions = mean_fn(Kin, Kcross, batch_nn_targets, **kwargs)
        scale = scale_fn(Kin, batch_nn_targets, **kwargs)
        variances = var_fn(Kin, Kcross, **kwargs)
        if target_mask is not None:
            predictions = predictions[:, target_mask]
            variances = variances[:, target_mask, target_mask]
        return -loss_fn(predictions, batch_targets, variances, scale, **loss_kwargs)

    return predict_and_loss_fn


class LossFn:
    def __init__(self, loss_fn: Callable, make_predict_and_loss_fn: Callable):
        self._fn = loss_fn
        self._make_predict_and_loss_fn = make_predict_and_loss_fn

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)

    def make_predict_and_loss_fn(self, *args, **kwargs):
        return self._make_predict_and_loss_fn(self._fn, *args, **kwargs)


cross_entropy_fn = LossFn(_cross_entropy_fn, make_raw_predict_and_loss_fn)
mse_fn = LossFn(_mse_fn, make_raw_predict_and_loss_fn)
lool_fn = LossFn(_lool_fn, make_var_predict_and_loss_fn)
lool_fn_unscaled = LossFn(_lool_fn_unscaled, make_var_predict_and_loss_fn)
pseudo_huber_fn = LossFn(_pseudo_huber_fn, make_raw_predict_and_loss_fn)
looph_fn = LossFn(_looph_fn, make_var_predict_and_loss_fn)
