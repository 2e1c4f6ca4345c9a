"""Fused hot-path entry points (one HIP launch from features + indices to mean/var).

These wrap ``mgp_posterior_{f32,f64}`` of the C ABI.  They are what the lazy tensor
handles of the ``hip`` backend call when a MuyGPyS-style caller runs
``make_predict_tensors -> kernel -> posterior_mean / posterior_variance`` (reference
call stack: gp/muygps.py:406-475, gp/kernels/matern.py:148-168, gp/muygps.py:164-259),
and what ``bench.py`` times.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Union

import torch

from . import _lib

Number = Union[int, float]


@dataclass
class KernelSpec:
    """Hyper-parameters of one local-GP model, in the reference's vocabulary.

    kernel: "rbf" | "matern05" | "matern15" | "matern25" | "maternInf"
        (_src/gp/kernels/numpy.py:12-31; Matern with fixed nu, gp/kernels/matern.py:61-81)
    metric: "l2" | "F2"  (gp/deformation/metric.py:237-265)
    length_scale: scalar -> Isotropy (isotropy.py:60-89); sequence of d -> Anisotropy
        (anisotropy.py:43-70)
    noise: scalar -> HomoscedasticNoise; 1-D tensor (n_train,) -> heteroscedastic table
        gathered with nn_indices (tensors/numpy.py:11-15); 2-D tensor (b, k) ->
        already gathered HeteroscedasticNoise tensor (noise/numpy.py:56-67)
    """

    kernel: str = "matern15"
    metric: str = "l2"
    length_scale: Union[Number, Sequence[Number], torch.Tensor] = 1.0
    noise: Union[Number, torch.Tensor] = 0.0

    def kernel_id(self) -> int:
        try:
            return _lib.KERNEL_IDS[self.kernel]
        except KeyError:
            raise ValueError(f"unknown kernel {self.kernel!r}; expected one of {sorted(_lib.KERNEL_IDS)}")

    def metric_id(self) -> int:
        try:
            return _lib.METRIC_IDS[self.metric]
        except KeyError:
            raise ValueError(f"unknown metric {self.metric!r}; expected 'l2' or 'F2'")


def _length_scale_tensor(ls, d: int, like: torch.Tensor) -> torch.Tensor:
    if isinstance(ls, torch.Tensor):
        t = ls.detach().to(device=like.device, dtype=like.dtype).reshape(-1)
    elif isinstance(ls, (int, float)):
        t = torch.tensor([float(ls)], device=like.device, dtype=like.dtype)
    else:
        t = torch.tensor([float(v) for v in ls], device=like.device, dtype=like.dtype)
    if t.numel() != 1 and t.numel() != d:
        raise ValueError(
            f"Difference tensor of shape (..., {d}) must have final dimension size of {t.numel()}"
        )
    return t.contiguous()


def _noise_args(noise, b: int, k: int, like: torch.Tensor):
    if isinstance(noise, torch.Tensor) and noise.ndim >= 1:
        _lib.require_cuda(noise)
        nz = noise.to(dtype=like.dtype).contiguous()
        if nz.ndim == 1:
            return _lib.NOISE_TABLE, 0.0, nz
        if nz.shape != (b, k):
            raise ValueError(f"heteroscedastic noise tensor must have shape {(b, k)}, got {tuple(nz.shape)}")
        return _lib.NOISE_BATCH, 0.0, nz
    return _lib.NOISE_SCALAR, float(noise), None


def posterior_mean_var(
    spec: KernelSpec,
    test_features: torch.Tensor,
    train_features: torch.Tensor,
    batch_indices: Optional[torch.Tensor],
    nn_indices: torch.Tensor,
    train_targets: torch.Tensor,
    want_ykinvy: bool = False,
    out_mean: Optional[torch.Tensor] = None,
    out_var: Optional[torch.Tensor] = None,
    info: Optional[torch.Tensor] = None,
):
    """Posterior mean and *unscaled* variance of every batch element, fused.

    Returns ``(mean, var)`` or ``(mean, var, ykinvy)``: mean is ``(b,)`` for 1-D targets
    and ``(b, R)`` for ``(n, R)`` targets (like _muygps_posterior_mean,
    _src/gp/muygps/numpy.py:17-41); var ``(b,)`` equals ``1 - Kcross K^-1 Kcross``
    (:44-67 with Kout = 1); ``ykinvy (b, R)`` holds ``y_r^T K^-1 y_r`` per neighbourhood
    (the summand of _analytic_scale_optim_unnormalized, scale/numpy.py:9-15).
    """
    _lib.require_cuda(test_features, train_features, batch_indices, nn_indices, train_targets)
    dtype = train_features.dtype
    if test_features.dtype != dtype or train_targets.dtype != dtype:
        raise TypeError("features and targets must share one float dtype")
    fq = test_features.contiguous()
    fn = train_features.contiguous()
    if fq.ndim == 1:
        fq = fq[:, None]
    if fn.ndim == 1:
        fn = fn[:, None]
    d = fn.shape[1]
    if fq.shape[1] != d:
        raise ValueError("test and train features differ in feature count")
    ni = nn_indices.to(torch.int64).contiguous()
    b, k = ni.shape
    bi = None if batch_indices is None else batch_indices.to(torch.int64).contiguous()
    if bi is not None and bi.shape != (b,):
        raise ValueError("batch_indices must have shape (batch_count,)")
    squeeze = train_targets.ndim == 1
    tg = (train_targets[:, None] if squeeze else train_targets).contiguous()
    R = tg.shape[1]
    ls = _length_scale_tensor(spec.length_scale, d, fn)
    mode, eps, nz = _noise_args(spec.noise, b, k, fn)

    mean = out_mean if out_mean is not None else torch.empty((b, R), device=fn.device, dtype=dtype)
    var = out_var if out_var is not None else torch.empty((b,), device=fn.device, dtype=dtype)
    yk = torch.empty((b, R), device=fn.device, dtype=dtype) if want_ykinvy else None
    rc = _lib.fn("posterior", dtype)(
        _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), R,
        mode, eps, _lib.ptr(nz), spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), ls.numel(),
        _lib.ptr(mean), _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), _lib.stream_ptr(),
    )
    _lib.check(rc, "mgp_posterior")
    mean_out = mean.reshape(b) if squeeze else mean.reshape(b, R)
    if want_ykinvy:
        return mean_out, var, (yk.reshape(b) if squeeze else yk)
    return mean_out, var
