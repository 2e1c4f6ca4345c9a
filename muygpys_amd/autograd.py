"""Differentiable fused posterior: ``mgp_posterior_*`` forward, ``mgp_posterior_backward_*`` backward.

The reference differentiates this path with torch autograd over its torch backend, keeping the
``(b,k,k,d)`` difference tensors alive (torch/muygps_layer.py:129-164; the training loop calls
``loss.sum().backward()`` at examples/muygps_torch.py:425-437).  Here ``torch.autograd.Function``
is only the plumbing that hands the cotangents to one HIP kernel per direction; nothing is
materialised and there is no CPU or eager fallback.
"""

from __future__ import annotations

from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from .fused import KernelSpec, _length_scale_tensor


def _column_sums(x2: torch.Tensor) -> torch.Tensor:
    """fp64 column sums of a (rows, cols) tensor through ``mgp_column_sums_*``."""
    return _lib.column_sums(x2.contiguous())


class _FusedPosterior(torch.autograd.Function):
    @staticmethod
    def forward(ctx, test_features, train_features, targets, length_scale, noise, batch_indices, nn_indices,
                kernel_id, metric_id, shared_table):
        fq, fn, tg = test_features.contiguous(), train_features.contiguous(), targets.contiguous()
        ls = length_scale.contiguous()
        b, k = nn_indices.shape
        d, R = fn.shape[1], tg.shape[1]
        if noise.ndim == 0:
            mode, eps, nz = _lib.NOISE_SCALAR, float(noise), None
        elif noise.ndim == 1:
            mode, eps, nz = _lib.NOISE_TABLE, 0.0, noise.contiguous()
        else:
            mode, eps, nz = _lib.NOISE_BATCH, 0.0, noise.contiguous()
        mean = torch.empty((b, R), device=fn.device, dtype=fn.dtype)
        var = torch.empty((b,), device=fn.device, dtype=fn.dtype)
        info = torch.zeros(1, device=fn.device, dtype=torch.int32)
        rc = _lib.fn("posterior", fn.dtype)(
            _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(batch_indices), _lib.ptr(nn_indices), b, k, _lib.ptr(tg), R,
            mode, eps, _lib.ptr(nz), kernel_id, metric_id, _lib.ptr(ls), ls.numel(),
            _lib.ptr(mean), _lib.ptr(var), None, _lib.ptr(info), _lib.stream_ptr(),
        )
        _lib.check(rc, "mgp_posterior")
        _lib.raise_if_not_spd(info, "posterior (autograd forward)")
        ctx.save_for_backward(fq, fn, tg, ls, noise, batch_indices, nn_indices)
        ctx.conf = (mode, eps, kernel_id, metric_id, shared_table)
        return mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_mean, grad_var):
        fq, fn, tg, ls, noise, bi, ni = ctx.saved_tensors
        mode, eps, kernel_id, metric_id, shared = ctx.conf
        need_q, need_x, need_t, need_l, need_n = ctx.needs_input_grad[:5]
        b, k = ni.shape
        d, R = fn.shape[1], tg.shape[1]
        dev, dt = fn.device, fn.dtype
        gm = None if grad_mean is None else grad_mean.to(dt).contiguous()
        gv = None if grad_var is None else grad_var.to(dt).contiguous()
        if gm is None and gv is None:
            return (None,) * 10
        nz = None if mode == _lib.NOISE_SCALAR else noise.contiguous()
        # one buffer when the query table is the training table (the LOOCV layout of MuyGPs_layer)
        g_x = torch.zeros_like(fn) if (need_x or (shared and need_q)) else None
        g_q = g_x if shared else (torch.zeros_like(fq) if need_q else None)
        g_t = torch.zeros_like(tg) if need_t else None
        g_l = torch.empty((b, ls.numel()), device=dev, dtype=dt) if need_l else None
        g_n = torch.empty((b, k), device=dev, dtype=dt) if need_n else None
        info = torch.zeros(1, device=dev, dtype=torch.int32)
        rc = _lib.fn("posterior_backward", dt)(
            _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), R,
            mode, eps, _lib.ptr(nz), kernel_id, metric_id, _lib.ptr(ls), ls.numel(),
            _lib.ptr(gm), _lib.ptr(gv), _lib.ptr(g_q), _lib.ptr(g_x), _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n),
            _lib.ptr(info), _lib.stream_ptr(),
        )
        _lib.check(rc, "mgp_posterior_backward")
        _lib.raise_if_not_spd(info, "posterior (autograd backward)")
        out_l = _column_sums(g_l).to(dt).reshape(ls.shape) if need_l else None
        out_n = None
        if need_n:
            if mode == _lib.NOISE_SCALAR:
                out_n = _column_sums(g_n.reshape(-1, 1)).to(dt).reshape(noise.shape)
            elif mode == _lib.NOISE_TABLE:
                out_n = torch.zeros_like(noise).index_add_(0, ni.reshape(-1), g_n.reshape(-1))
            else:
                out_n = g_n
        if shared:
            # both roles accumulate in g_x; hand it to whichever leaf asked (autograd adds them anyway)
            return (g_x if need_q and not need_x else None, g_x if need_x else None, g_t, out_l, out_n,
                    None, None, None, None, None)
        return (g_q if need_q else None, g_x if need_x else None, g_t, out_l, out_n, None, None, None, None, None)


def _as_param(v, like: torch.Tensor) -> torch.Tensor:
    if isinstance(v, torch.Tensor):
        return v.to(device=like.device, dtype=like.dtype)
    return torch.as_tensor(v, device=like.device, dtype=like.dtype)


def posterior(
    spec: KernelSpec,
    test_features: torch.Tensor,
    train_features: torch.Tensor,
    batch_indices: Optional[torch.Tensor],
    nn_indices: torch.Tensor,
    train_targets: torch.Tensor,
):
    """Differentiable ``(mean, var)`` of every neighbourhood (unscaled variance ``1 - c^T K^-1 c``).

    Same arguments and results as :func:`muygpys_amd.fused.posterior_mean_var`; gradients flow to
    ``test_features``, ``train_features``, ``train_targets`` and to ``spec.length_scale`` /
    ``spec.noise`` when those are tensors that require grad.  Passing the *same tensor* as
    ``test_features`` and ``train_features`` (``MuyGPs_layer.forward``: crosswise_tensor(x, x, ...),
    torch/muygps_layer.py:146-155) accumulates both roles into one gradient buffer."""
    _lib.require_cuda(test_features, train_features, batch_indices, nn_indices, train_targets)
    dtype = train_features.dtype
    if test_features.dtype != dtype or train_targets.dtype != dtype:
        raise TypeError("features and targets must share one float dtype")
    shared = test_features is train_features
    fq = test_features[:, None] if test_features.ndim == 1 else test_features
    fn = fq if shared else (train_features[:, None] if train_features.ndim == 1 else train_features)
    d = fn.shape[1]
    if fq.shape[1] != d:
        raise ValueError("test and train features differ in feature count")
    ni = nn_indices.to(torch.int64).contiguous()
    b, k = ni.shape
    kmax = _lib.load().mgp_max_nn_count_backward(4 if dtype == torch.float32 else 8)
    if k > kmax:
        raise ValueError(f"nn_count {k} exceeds the differentiable kernel's limit {kmax} for {dtype}")
    if batch_indices is None:
        bi = torch.arange(b, device=ni.device, dtype=torch.int64)
    else:
        bi = batch_indices.to(torch.int64).contiguous()
    squeeze = train_targets.ndim == 1
    tg = train_targets[:, None] if squeeze else train_targets
    if isinstance(spec.length_scale, torch.Tensor):
        ls = spec.length_scale.to(device=fn.device, dtype=dtype).reshape(-1)
        if ls.numel() not in (1, d):
            raise ValueError(
                f"Difference tensor of shape (..., {d}) must have final dimension size of {ls.numel()}"
            )
    else:
        ls = _length_scale_tensor(spec.length_scale, d, fn)
    noise = _as_param(spec.noise, fn)
    if noise.ndim == 2 and noise.shape != (b, k):
        raise ValueError(f"heteroscedastic noise tensor must have shape {(b, k)}, got {tuple(noise.shape)}")
    if noise.ndim == 1 and noise.shape[0] != fn.shape[0]:
        raise ValueError(
            f"per-training-point noise table holds {noise.shape[0]} entries for {fn.shape[0]} training points"
        )
    mean, var = _FusedPosterior.apply(fq, fn, tg, ls, noise, bi, ni, spec.kernel_id(), spec.metric_id(), shared)
    return (mean.reshape(b) if squeeze else mean), var
