"""hip implementation of the analytic scale family (reference: _src/optimize/scale/numpy.py)."""

from __future__ import annotations

import torch

from muygpys_amd import _lib
from muygpys_amd._src.gp.muygps.hip import _solve


def _ykinvy_sums(Kin, Y):
    """sum_b y_r^T Kin^-1 y_r per response column, as a device float64 vector (R,)."""
    _, _, yk, _ = _solve(Kin, None, Y, want=("ykinvy",))
    b, R = yk.shape
    out = torch.empty((R,), device=yk.device, dtype=torch.float64)
    rc = _lib.fn("column_sums", yk.dtype)(_lib.ptr(yk), b, R, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "mgp_column_sums")
    return out


def _analytic_scale_optim_unnormalized(Kin, nn_targets, **kwargs):
    """numpy.py:9-15: sum over batch AND response columns of y^T Kin^-1 y (0-d tensor)."""
    _lib.require_cuda(Kin, nn_targets)
    Y = nn_targets if nn_targets.ndim == 3 else nn_targets[:, :, None]
    return _ykinvy_sums(Kin, Y).sum().to(Kin.dtype)


def _analytic_scale_optim(Kin, nn_targets, batch_dim_count: int = 1, **kwargs):
    """numpy.py:18-34: / (batch_size * nn_count); like the reference, R > 1 is rejected
    (its reshape to (b, k, 1) raises ValueError)."""
    _lib.require_cuda(Kin, nn_targets)
    b, k, _ = Kin.shape
    if nn_targets.numel() != b * k:
        raise ValueError(
            f"cannot reshape array of size {nn_targets.numel()} into shape ({b},{k},1)"
        )
    return _analytic_scale_optim_unnormalized(Kin, nn_targets.reshape(b, k, 1)) / (b * k)
