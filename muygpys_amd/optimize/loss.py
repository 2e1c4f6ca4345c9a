"""Loss objects of the optimisation layer.

Contract (src/MuyGPyS/optimize/loss.py:26-396): a loss object is callable like its backend function and
offers ``make_predict_and_loss_fn(mean_fn, var_fn, scale_fn, batch_nn_targets, batch_targets, **loss_kwargs)``,
which returns ``f(Kin, Kcross, **kwargs) -> -loss``.  Minus, because the Bayesian driver maximises
(loss.py:94,174).

Here the catalogue is one table.  Each entry names the backend function and which posterior
quantities its objective needs:

* ``"mean"`` -- the loss compares posterior means with the batch targets (mse, pseudo-Huber,
  cross-entropy);
* ``"mean+var"`` -- it also takes the unscaled posterior variance and the analytic sigma^2 (lool,
  looph).  The three are requested in the reference's order (mean, scale, variance); on lazy handles
  they are one fused launch (``muygpys_amd.lazy_eval``).

One closure builder serves both kinds.
"""

from __future__ import annotations

from typing import Callable, Dict, Tuple

from muygpys_amd._src.optimize import loss as _backend

_CATALOGUE: Dict[str, Tuple[str, str]] = {
    # public name        backend function        what the objective evaluates
    "cross_entropy_fn": ("_cross_entropy_fn", "mean"),
    "mse_fn": ("_mse_fn", "mean"),
    "pseudo_huber_fn": ("_pseudo_huber_fn", "mean"),
    "lool_fn": ("_lool_fn", "mean+var"),
    "lool_fn_unscaled": ("_lool_fn_unscaled", "mean+var"),
    "looph_fn": ("_looph_fn", "mean+var"),
}


def _objective_closure(needs: str, loss: Callable, mean_fn, var_fn, scale_fn, batch_nn_targets, batch_targets,
                       target_mask=None, **loss_kwargs) -> Callable:
    """``f(Kin, Kcross, **kwargs) -> -loss`` for one batch.  ``target_mask`` selects response columns of
    the mean (and the matching diagonal block of a full covariance)."""
    with_variance = needs == "mean+var"

    def negative_loss(Kin, Kcross, *unused, **kwargs):
        operands = [mean_fn(Kin, Kcross, batch_nn_targets, **kwargs), batch_targets]
        if with_variance:
            scale = scale_fn(Kin, batch_nn_targets, **kwargs)
            operands += [var_fn(Kin, Kcross, **kwargs), scale]
        if target_mask is not None:
            operands[0] = operands[0][:, target_mask]
            if with_variance:
                operands[2] = operands[2][:, target_mask, target_mask]
        return -loss(*operands, **loss_kwargs)

    return negative_loss


def make_raw_predict_and_loss_fn(loss_fn: Callable, *args, **kwargs) -> Callable:
    """The reference's builder name for mean-only losses (loss.py:26-96)."""
    return _objective_closure("mean", loss_fn, *args, **kwargs)


def make_var_predict_and_loss_fn(loss_fn: Callable, *args, **kwargs) -> Callable:
    """The reference's builder name for losses that take variance and scale (loss.py:99-178)."""
    return _objective_closure("mean+var", loss_fn, *args, **kwargs)


class LossFn:
    """A backend loss plus the recipe that evaluates it inside an objective.  The second argument is a
    builder ``(loss_fn, mean_fn, var_fn, scale_fn, batch_nn_targets, batch_targets, ...) -> closure``
    (the two above, or a caller's own)."""

    def __init__(self, loss_fn: Callable, make_predict_and_loss_fn: Callable):
        self._fn = loss_fn
        self._make_predict_and_loss_fn = make_predict_and_loss_fn

    def __call__(self, *args, **kwargs):
        return self._fn(*args, **kwargs)

    def make_predict_and_loss_fn(self, *args, **kwargs) -> Callable:
        return self._make_predict_and_loss_fn(self._fn, *args, **kwargs)


_BUILDERS = {"mean": make_raw_predict_and_loss_fn, "mean+var": make_var_predict_and_loss_fn}
globals().update(
    {name: LossFn(getattr(_backend, fn), _BUILDERS[needs]) for name, (fn, needs) in _CATALOGUE.items()}
)
# (spelled out for readers and static tools)
cross_entropy_fn: LossFn
mse_fn: LossFn
pseudo_huber_fn: LossFn
lool_fn: LossFn
lool_fn_unscaled: LossFn
looph_fn: LossFn
__all__ = ["LossFn", "make_raw_predict_and_loss_fn", "make_var_predict_and_loss_fn", *_CATALOGUE]
