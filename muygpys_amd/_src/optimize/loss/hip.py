"""hip implementation of the loss family (reference: _src/optimize/loss/numpy.py).

One fp64 reduction kernel (``mgp_loss_sums_*``) produces every sum the losses need; each
function below picks its entry.  Results are 0-d device tensors of the input dtype, like
the torch backend returns.
"""

from __future__ import annotations

import torch

from muygpys_amd import _lib


def _sums(predictions, targets, variances=None, scale=None, huber_delta=1.5, looph_delta=3.0):
    _lib.require_cuda(predictions, targets, variances)
    p = predictions.contiguous()
    t = targets.to(p.dtype).contiguous()
    if p.shape != t.shape:
        if p.numel() != t.numel():
            raise ValueError(f"predictions {tuple(p.shape)} and targets {tuple(t.shape)} do not conform")
        t = t.reshape(p.shape)
    v = None
    if variances is not None:
        if variances.ndim != 1:
            raise NotImplementedError("full-covariance variances are outside the hip hot path")
        v = variances.to(p.dtype).contiguous()
        if v.numel() != p.numel():
            raise ValueError("variances must have one entry per prediction (1-D responses)")
    s = None
    if scale is not None:
        s = (scale.detach() if isinstance(scale, torch.Tensor) else torch.tensor(float(scale)))
        s = s.to(device=p.device, dtype=torch.float64).reshape(-1)[:1].contiguous()
    out = torch.empty(6, device=p.device, dtype=torch.float64)
    rc = _lib.fn("loss_sums", p.dtype)(
        _lib.ptr(p), _lib.ptr(t), _lib.ptr(v), p.numel(), _lib.ptr(s), float(huber_delta), float(looph_delta),
        _lib.ptr(out), _lib.stream_ptr(),
    )
    _lib.check(rc, "mgp_loss_sums")
    return out


def _cross_entropy_fn(predictions, targets, **kwargs):
    """numpy.py:12-19 calls sklearn.metrics.log_loss (classification only): out of scope."""
    raise NotImplementedError("The hip backend does not implement the cross-entropy loss.")


def _mse_fn(predictions, targets, **kwargs):
    """numpy.py:22-31."""
    return (_sums(predictions, targets)[0] / predictions.numel()).to(predictions.dtype)


def _lool_fn_unscaled(predictions, targets, variances, **kwargs):
    """numpy.py:34-51 (1-D variance branch)."""
    return _sums(predictions, targets, variances)[1].to(predictions.dtype)


def _lool_fn(predictions, targets, variances, scale, **kwargs):
    """numpy.py:54-61."""
    return _sums(predictions, targets, variances, scale=scale)[1].to(predictions.dtype)


def _pseudo_huber_fn(predictions, targets, boundary_scale: float = 1.5, **kwargs):
    """numpy.py:64-72."""
    return _sums(predictions, targets, huber_delta=boundary_scale)[2].to(predictions.dtype)


def _looph_fn(predictions, targets, variances, scale, boundary_scale: float = 3.0, **kwargs):
    """numpy.py:75-117."""
    if variances.ndim != 1:
        raise ValueError("looph does not yet support multivariate inference")
    return _sums(predictions, targets, variances, scale=scale, looph_delta=boundary_scale)[3].to(predictions.dtype)
