"""hip implementation of the tensor family (reference: src/MuyGPyS/_src/gp/tensors/numpy.py).

Called directly these are materialising kernels for API parity: each returns a fresh device
tensor like the numpy backend does.  With ``config.state.lazy_tensors`` (what
``integration.install()`` switches on) ``_pairwise_tensor`` / ``_crosswise_tensor`` return the
light handles of ``muygpys_amd.lazy`` instead and ``_l2`` / ``_F2`` decorate them, so that the
functor layer above -- this package's or the reference's own -- ends in one fused launch.
"""

from __future__ import annotations

from typing import Tuple

import torch

from muygpys_amd import _lib, lazy
from muygpys_amd.config import config as _config


def _feat2d(x: torch.Tensor) -> torch.Tensor:
    return (x[:, None] if x.ndim == 1 else x).contiguous()


def _idx(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.int64).contiguous()


def _make_heteroscedastic_tensor(measurement_noise, batch_nn_indices):
    """numpy.py:11-15."""
    _lib.require_cuda(measurement_noise, batch_nn_indices)
    return measurement_noise[batch_nn_indices]


def _batch_features_tensor(features, batch_indices):
    """numpy.py:40-44."""
    _lib.require_cuda(features, batch_indices)
    return features[batch_indices]


def _crosswise_tensor(data, nn_data, data_indices, nn_indices):
    """numpy.py:47-58: (b, k, d) differences query - neighbour; 1-D data -> (b, k, 1)."""
    if _config.state.lazy_tensors:
        _lib.require_cuda(data, nn_data, data_indices, nn_indices)
        return lazy.LazyDiffs("crosswise", None, False, nn_data, nn_indices, data, data_indices)
    return _crosswise_tensor_now(data, nn_data, data_indices, nn_indices)


def _crosswise_tensor_now(data, nn_data, data_indices, nn_indices):
    _lib.require_cuda(data, nn_data, data_indices, nn_indices)
    fq, fn = _feat2d(data), _feat2d(nn_data)
    bi, ni = _idx(data_indices), _idx(nn_indices)
    b, k = ni.shape
    d = fn.shape[1]
    out = torch.empty((b, k, d), device=fn.device, dtype=fn.dtype)
    rc = _lib.fn("crosswise_diffs", fn.dtype)(
        _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(out), _lib.stream_ptr()
    )
    _lib.check(rc, "mgp_crosswise_diffs")
    return out


def _pairwise_tensor(data, nn_indices):
    """numpy.py:61-69: (b, k, k, d) with [b,i,j,:] = x_i - x_j."""
    if _config.state.lazy_tensors:
        _lib.require_cuda(data, nn_indices)
        return lazy.LazyDiffs("pairwise", None, False, data, nn_indices)
    return _pairwise_tensor_now(data, nn_indices)


def _pairwise_tensor_now(data, nn_indices):
    _lib.require_cuda(data, nn_indices)
    f = _feat2d(data)
    ni = _idx(nn_indices)
    b, k = ni.shape
    d = f.shape[1]
    out = torch.empty((b, k, k, d), device=f.device, dtype=f.dtype)
    rc = _lib.fn("pairwise_diffs", f.dtype)(_lib.ptr(f), d, _lib.ptr(ni), b, k, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "mgp_pairwise_diffs")
    return out


def _crosswise_differences(locations, points):
    """numpy.py:72-75: locations (b, d), points (b, k, d)."""
    _lib.require_cuda(locations, points)
    return locations[:, None, :] - points


def _pairwise_differences(points):
    """numpy.py:78-86."""
    _lib.require_cuda(points)
    if points.ndim == 1:
        return (points[:, None] - points[None, :])[:, :, None]
    if points.ndim == 2:
        return points[:, None, :] - points[None, :, :]
    if points.ndim == 3:
        return points[:, :, None, :] - points[:, None, :, :]
    raise ValueError(f"points shape {tuple(points.shape)} is not supported.")


def _reduce(diffs, metric_id, length_scale=None):
    _lib.require_cuda(diffs)
    x = diffs.contiguous()
    d = x.shape[-1]
    n = x.numel() // d if d else 0
    out = torch.empty(x.shape[:-1], device=x.device, dtype=x.dtype)
    ls = None if length_scale is None else length_scale.to(device=x.device, dtype=x.dtype).contiguous()
    rc = _lib.fn("reduce_diffs", x.dtype)(_lib.ptr(x), n, d, _lib.ptr(ls), metric_id, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "mgp_reduce_diffs")
    return out


def _metric(diffs, name: str):
    if isinstance(diffs, lazy.LazyDiffs):
        if not diffs.reduced:
            return diffs.reduce(name)
        diffs = diffs.materialize()
    lsv = None
    return _reduce(diffs, _lib.METRIC_IDS[name], lsv)


def _F2(diffs):
    """numpy.py:89-90."""
    return _metric(diffs, "F2")


def _l2(diffs):
    """numpy.py:93-94."""
    return _metric(diffs, "l2")


def _crosswise_distances(data, nn_data, data_indices, nn_indices, metric: str):
    """T1+T3 fused -- what Isotropy.crosswise_tensor returns (isotropy.py:121-161)."""
    _lib.require_cuda(data, nn_data, data_indices, nn_indices)
    fq, fn = _feat2d(data), _feat2d(nn_data)
    bi, ni = _idx(data_indices), _idx(nn_indices)
    b, k = ni.shape
    out = torch.empty((b, k), device=fn.device, dtype=fn.dtype)
    rc = _lib.fn("crosswise_dists", fn.dtype)(
        _lib.ptr(fq), _lib.ptr(fn), fn.shape[1], _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.METRIC_IDS[metric],
        _lib.ptr(out), _lib.stream_ptr(),
    )
    _lib.check(rc, "mgp_crosswise_dists")
    return out


def _pairwise_distances(data, nn_indices, metric: str):
    """T2+T3 fused -- what Isotropy.pairwise_tensor returns (isotropy.py:92-118); the
    (b, k, k, d) difference tensor is never formed."""
    _lib.require_cuda(data, nn_indices)
    f = _feat2d(data)
    ni = _idx(nn_indices)
    b, k = ni.shape
    out = torch.empty((b, k, k), device=f.device, dtype=f.dtype)
    rc = _lib.fn("pairwise_dists", f.dtype)(
        _lib.ptr(f), f.shape[1], _lib.ptr(ni), b, k, _lib.METRIC_IDS[metric], _lib.ptr(out), _lib.stream_ptr()
    )
    _lib.check(rc, "mgp_pairwise_dists")
    return out


def _fast_nn_update(train_nn_indices):
    """numpy.py:97-108."""
    _lib.require_cuda(train_nn_indices)
    n = train_nn_indices.shape[0]
    me = torch.arange(n, device=train_nn_indices.device, dtype=train_nn_indices.dtype)
    return torch.cat((me[:, None], train_nn_indices[:, :-1]), dim=1)


def _make_fast_predict_tensors(batch_nn_indices, train_features, train_targets) -> Tuple[torch.Tensor, torch.Tensor]:
    """numpy.py:18-37."""
    idx_fast = _fast_nn_update(batch_nn_indices)
    return _pairwise_tensor(train_features, idx_fast), train_targets[idx_fast]
