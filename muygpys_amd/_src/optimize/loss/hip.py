"""hip implementation of the loss family (reference: _src/optimize/loss/numpy.py).

One fp64 reduction kernel (``mgp_loss_sums_*``) produces every sum the losses need; each
function below picks its entry.  Results are 0-d device tensors of the input dtype, like
the torch backend returns.
"""

from __future__ import annotations

import torch

from muygpys_amd import _lib


def _sums_and_count(predictions, targets, variances=None, scale=None, huber_delta=1.5, looph_delta=3.0):
    """(the six fp64 sums, element count): both global when the batch is sharded over ranks."""
    _lib.require_cuda(predictions, targets, variances)
    p = predictions.contiguous()
    t = targets.to(p.dtype).contiguous()
    if p.shape != t.shape:
        if p.numel() != t.numel():
            raise ValueError(f"predictions {tuple(p.shape)} and targets {tuple(t.shape)} do not conform")
        t = t.reshape(p.shape)
    v = None
    if variances is not None:
        if variances.ndim == 3:
            raise NotImplementedError("full-covariance variances are outside the hip hot path")
        v = variances.to(p.dtype).contiguous()
        if v.numel() != p.numel():
            raise ValueError("variances must have one entry per prediction (1-D responses)")
    s = None
    if scale is not None:
        s = scale.detach() if isinstance(scale, torch.Tensor) else torch.tensor(float(scale), dtype=torch.float64)
        s = s.to(device=p.device, dtype=torch.float64).reshape(-1)[:1].contiguous()
    out = _lib.loss_sums(p, t, v, s, huber_delta, looph_delta)
    from muygpys_amd import distributed as _D

    if _D.reductions_active():
        # sharded batch: the sums (and, for mse, the element count) are global, like the reference's
        # mpi backend (_src/optimize/loss/mpi.py:20-104)
        tot = torch.cat([out, torch.tensor([float(p.numel())], device=out.device, dtype=torch.float64)])
        _D.reduce_if_sharded_(tot)
        return tot[:6], tot[6]
    return out, p.numel()


def _sums(*args, **kwargs):
    return _sums_and_count(*args, **kwargs)[0]


def _wants_grad(*xs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


class _LoolSum(torch.autograd.Function):
    """sum(r^2 / (s v) + log(s v)) with the HIP reduction as forward and the closed-form elementwise
    cotangents as backward, so a deep-kernel loop (examples/muygps_torch.py:425-437) can call
    ``loss.backward()`` on the hip backend's own loss."""

    @staticmethod
    def forward(ctx, predictions, targets, variances, scale):
        ctx.save_for_backward(predictions, targets, variances)
        ctx.scale = 1.0 if scale is None else float(scale)
        return _sums(predictions.detach(), targets, variances.detach(), scale=scale)[1].to(predictions.dtype)

    @staticmethod
    def backward(ctx, g):
        p, t, v = ctx.saved_tensors
        r = p - t.to(p.dtype).reshape(p.shape)
        vv = v.reshape(p.shape) * ctx.scale
        gp = g * 2.0 * r / vv if ctx.needs_input_grad[0] else None
        gv = (g * (1.0 / vv - (r / vv) ** 2) * ctx.scale).reshape(v.shape) if ctx.needs_input_grad[2] else None
        return gp, None, gv, None


class _MseMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predictions, targets):
        ctx.save_for_backward(predictions, targets)
        sums, count = _sums_and_count(predictions.detach(), targets)
        ctx.count = float(count)  # the GLOBAL element count under sharded reductions
        return (sums[0] / count).to(predictions.dtype)

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return g * 2.0 * (p - t.to(p.dtype).reshape(p.shape)) / ctx.count, None


def _cross_entropy_fn(predictions, targets, **kwargs):
    """numpy.py:12-19 calls sklearn.metrics.log_loss (classification only): out of scope."""
    raise NotImplementedError("The hip backend does not implement the cross-entropy loss.")


def _mse_fn(predictions, targets, **kwargs):
    """numpy.py:22-31."""
    if _wants_grad(predictions):
        return _MseMean.apply(predictions, targets)
    sums, count = _sums_and_count(predictions, targets)
    return (sums[0] / count).to(predictions.dtype)


def _lool_fn_unscaled(predictions, targets, variances, **kwargs):
    """numpy.py:34-51 (1-D variance branch; elementwise for same-shape 2-D, torch.py:62-65)."""
    if _wants_grad(predictions, variances):
        return _LoolSum.apply(predictions, targets, variances, None)
    return _sums(predictions, targets, variances)[1].to(predictions.dtype)


def _lool_fn(predictions, targets, variances, scale, **kwargs):
    """numpy.py:54-61."""
    if _wants_grad(predictions, variances):
        sv = float(scale.detach().reshape(-1)[0]) if isinstance(scale, torch.Tensor) else float(scale)
        return _LoolSum.apply(predictions, targets, variances, sv)
    return _sums(predictions, targets, variances, scale=scale)[1].to(predictions.dtype)


def _pseudo_huber_fn(predictions, targets, boundary_scale: float = 1.5, **kwargs):
    """numpy.py:64-72."""
    return _sums(predictions, targets, huber_delta=boundary_scale)[2].to(predictions.dtype)


def _looph_fn(predictions, targets, variances, scale, boundary_scale: float = 3.0, **kwargs):
    """numpy.py:75-117."""
    if variances.ndim != 1:
        raise ValueError("looph does not yet support multivariate inference")
    return _sums(predictions, targets, variances, scale=scale, looph_delta=boundary_scale)[3].to(predictions.dtype)
