"""``mm`` facade of the hip backend: torch tensors resident on the ROCm device.

Exports the 51 names of src/MuyGPyS/_src/math/__init__.py:8-109 with the reference's dtype
rules (numpy.py:92-104, meta.py:9-43): float constructors default to ``ftype`` (float64
unless MUYGPYS_FTYPE=32), ``iarray``/``arange`` to int64, ``assign`` copies, and
``parameter`` is the identity (the hip path has no autograd, so unlike the torch backend's
``nn.Parameter`` it keeps plain values).  Tensors are created on the current ROCm device.
"""

from __future__ import annotations

import torch
from torch import (  # noqa: F401
    Tensor as ndarray,
    all,
    allclose,
    atleast_1d,
    atleast_2d,
    corrcoef,
    cov,
    exp,
    float32,
    float64,
    inf,
    int32,
    int64,
    isclose,
    linalg,
    log,
    logical_or,
    mean,
    median,
    outer,
    prod,
    reshape,
    sqrt,
    squeeze,
    tile,
    unique,
    vstack,
    where,
)
from torch import div as divide  # noqa: F401
from torch import repeat_interleave as repeat  # noqa: F401
from torch.linalg import cholesky  # noqa: F401

from muygpys_amd.config import config

ftype = float32 if config.state.low_precision() else float64
itype = int64


def _device():
    """The current ROCm device; there is no host fallback (ValueError without a device, like an
    unavailable backend in the reference, _src/config.py:230-243)."""
    config.require_device()
    return torch.device("cuda", torch.cuda.current_device())


def _typed(dtype, fn):
    def typed_fn(*args, **kwargs):
        kwargs.setdefault("dtype", dtype)
        kwargs.setdefault("device", _device())
        return fn(*args, **kwargs)

    return typed_fn


def _as_tensor(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(device=_device(), dtype=dtype)
    return torch.as_tensor(x, dtype=dtype, device=_device())


class TableTensor(torch.Tensor):
    """What the float constructors of this facade return: a ``torch.Tensor`` (``isinstance`` holds, same
    storage, every torch function works) whose gather by a 2-D integer index tensor --
    ``train_targets[batch_nn_indices]``, the one line of the reference's functor layer that is NOT a
    backend function (gp/muygps.py:474,543) -- yields the lazy handle :class:`muygpys_amd.lazy.LazyTargets`
    while ``config.state.lazy_tensors`` is on (``integration.install()``).  The handle remembers
    (table, indices); the fused launch then reads the responses with the neighbour rows from the
    prepared table (``mgp_posterior_packed_*``) instead of from a materialised ``(b, k, R)`` copy.  Any
    other use of the handle materialises it.  A plain tensor is wrapped, without a copy, by
    ``muygpys_amd.integration.table(t)``."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """Only the table itself carries the lazy gather: what a torch function computes FROM it is a plain
        ``torch.Tensor`` (the default would hand the subclass on to every result -- ``features * 2``,
        ``targets.mean()`` ... -- which would then turn their own 2-D integer gathers into handles and pay this
        dispatch on every later op; round-3 advisor finding)."""
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    def __getitem__(self, item):
        if (
            config.state.lazy_tensors and isinstance(item, torch.Tensor) and item.ndim == 2
            and item.dtype in (torch.int64, torch.int32) and self.ndim in (1, 2) and self.is_floating_point()
        ):
            from muygpys_amd import lazy

            return lazy.LazyTargets(self.as_subclass(torch.Tensor), item)
        return torch.Tensor.__getitem__(self.as_subclass(torch.Tensor), item)


def farray(x, **kwargs):
    return _as_tensor(x, kwargs.get("dtype", ftype)).as_subclass(TableTensor)


def iarray(x, **kwargs):
    return _as_tensor(x, kwargs.get("dtype", itype))


array = farray
arange = _typed(itype, torch.arange)
eye = _typed(ftype, torch.eye)
full = _typed(ftype, torch.full)
linspace = _typed(ftype, torch.linspace)
ones = _typed(ftype, torch.ones)
zeros = _typed(ftype, torch.zeros)
diagonal = torch.diagonal


def _axis_kw(fn):
    """numpy spells it ``axis``; torch ``dim`` (reference: meta.py wrap_torch_signatures)."""

    def wrapped(x, *args, axis=None, **kwargs):
        if axis is not None:
            kwargs["dim"] = axis
        return fn(x, *args, **kwargs)

    return wrapped


argmax = _axis_kw(torch.argmax)
max = _axis_kw(torch.max)
min = _axis_kw(torch.min)
sum = _axis_kw(torch.sum)


def assign(x, y, *slices):
    """Copy-on-write assignment (numpy.py:86-89)."""
    ret = torch.clone(x)
    ret[slices] = y
    return ret


def parameter(x):
    return x
