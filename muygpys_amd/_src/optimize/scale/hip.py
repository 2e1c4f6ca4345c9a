"""hip implementation of the analytic scale family (reference: _src/optimize/scale/numpy.py)."""

from __future__ import annotations

import torch

from muygpys_amd import _lib, lazy, lazy_eval
from muygpys_amd._src.gp.muygps.hip import _solve


def _ykinvy_sums(Kin, Y):
    """sum_b y_r^T Kin^-1 y_r per response column, as a device float64 vector (R,)."""
    _, _, yk, _ = _solve(Kin, None, Y, want=("ykinvy",))
    return _lib.column_sums(yk.contiguous())


def _analytic_scale_optim_unnormalized(Kin, nn_targets, **kwargs):
    """numpy.py:9-15: sum over batch AND response columns of y^T Kin^-1 y (0-d tensor)."""
    Kin, nn_targets = lazy.force(Kin), lazy.force(nn_targets)
    _lib.require_cuda(Kin, nn_targets)
    Y = nn_targets if nn_targets.ndim == 3 else nn_targets[:, :, None]
    from muygpys_amd import distributed as _D

    return _D.reduce_if_sharded_(_ykinvy_sums(Kin, Y).sum().reshape(1))[0].to(Kin.dtype)


def _analytic_scale_optim(Kin, nn_targets, batch_dim_count: int = 1, **kwargs):
    """numpy.py:18-34: / (batch_size * nn_count); like the reference, R > 1 is rejected
    (its reshape to (b, k, 1) raises ValueError)."""
    if isinstance(Kin, lazy.LazyCov):
        out = lazy_eval.analytic_scale(Kin, nn_targets)
        if out is not None:
            return out
    Kin, nn_targets = lazy.force(Kin), lazy.force(nn_targets)
    _lib.require_cuda(Kin, nn_targets)
    b, k, _ = Kin.shape
    if nn_targets.numel() != b * k:
        raise ValueError(
            f"cannot reshape array of size {nn_targets.numel()} into shape ({b},{k},1)"
        )
    from muygpys_amd import distributed as _D

    if _D.reductions_active():
        # sharded batch: global sum and global batch count (_src/optimize/scale/mpi.py:16-37)
        Y = nn_targets.reshape(b, k, 1)
        tot = torch.cat([_ykinvy_sums(Kin, Y).sum().reshape(1),
                         torch.tensor([float(b)], device=Kin.device, dtype=torch.float64)])
        _D.reduce_if_sharded_(tot)
        return (tot[0] / (tot[1] * k)).to(Kin.dtype)
    return _analytic_scale_optim_unnormalized(Kin, nn_targets.reshape(b, k, 1)) / (b * k)
