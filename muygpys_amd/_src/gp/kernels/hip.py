"""hip implementation of the kernel family (reference: src/MuyGPyS/_src/gp/kernels/numpy.py)."""

from __future__ import annotations

import torch

from muygpys_amd import _lib, lazy


def _apply(dists, kernel: str, in_scale: float = 1.0):
    if isinstance(dists, lazy.LazyDiffs):
        if dists.reduced and in_scale == 1.0:
            return lazy.LazyCov(dists, kernel)  # evaluated by the fused launch (or on demand)
        dists = dists.materialize()
    _lib.require_cuda(dists)
    x = dists.contiguous()
    out = torch.empty_like(x)
    rc = _lib.fn("kernel_apply", x.dtype)(
        _lib.ptr(x), x.numel(), _lib.KERNEL_IDS[kernel], float(in_scale), _lib.ptr(out), _lib.stream_ptr()
    )
    _lib.check(rc, "mgp_kernel_apply")
    return out


def _rbf_fn(squared_dists, **kwargs):
    """numpy.py:12-13."""
    return _apply(squared_dists, "rbf")


def _matern_05_fn(dists, **kwargs):
    """numpy.py:16-17."""
    return _apply(dists, "matern05")


def _matern_15_fn(dists, **kwargs):
    """numpy.py:20-22."""
    return _apply(dists, "matern15")


def _matern_25_fn(dists, **kwargs):
    """numpy.py:25-27."""
    return _apply(dists, "matern25")


def _matern_inf_fn(dists, **kwargs):
    """numpy.py:30-31."""
    return _apply(dists, "maternInf")


def _matern_gen_fn(dists, smoothness, **kwargs):
    """numpy.py:34-43: general smoothness, 2^(1-nu)/Gamma(nu) (sqrt(2 nu) r)^nu K_nu(sqrt(2 nu) r)
    with zeros replaced by eps -- ``mgp_matern_gen_*`` evaluates the modified Bessel function on the
    device in fp64 (the reference calls scipy.special.kv).  The input is left untouched (the
    reference overwrites it)."""
    nu = float(smoothness.detach().reshape(-1)[0]) if isinstance(smoothness, torch.Tensor) else float(smoothness)
    if not nu > 0.0:
        raise ValueError(f"Matern smoothness must be positive, got {nu}")
    if isinstance(dists, lazy.LazyDiffs):
        if dists.reduced:
            # a handle: the fused launch evaluates the kernel itself where it can (fp32 tables,
            # mgp_posterior_gen_*); where it cannot, the handle materialises through this function
            return lazy.LazyCov(dists, "matern_gen", smoothness=nu)
        dists = dists.scaled_distances()
    _lib.require_cuda(dists)
    x = dists.contiguous()
    out = torch.empty_like(x)
    rc = _lib.fn("matern_gen", x.dtype)(_lib.ptr(x), x.numel(), 1.0, nu, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "mgp_matern_gen")
    return out
