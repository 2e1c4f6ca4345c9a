"""hip implementation of the kernel family (reference: src/MuyGPyS/_src/gp/kernels/numpy.py)."""

from __future__ import annotations

import torch

from muygpys_amd import _lib


def _apply(dists, kernel: str, in_scale: float = 1.0):
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
    """numpy.py:34-43 needs scipy's modified Bessel function kv; like the reference's torch
    backend (torch.py:26-32) the hip backend does not provide it."""
    raise NotImplementedError(
        'The hip backend does not implement the general-smoothness Matern kernel (scipy.special.kv); '
        "fix smoothness to one of 0.5, 1.5, 2.5, inf."
    )
