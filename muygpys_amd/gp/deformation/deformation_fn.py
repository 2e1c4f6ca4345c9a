"""Deformation functors: Isotropy (scalar length scale on distances) and Anisotropy (one
length scale per feature, on differences).  Reference contract:
src/MuyGPyS/gp/deformation/{deformation_fn,isotropy,anisotropy}.py.

Which tensor a deformation materialises decides what the kernel sees: Isotropy hands
DISTANCES (b,k[,k]) to the kernel (isotropy.py:92-161), Anisotropy raw DIFFERENCES
(b,k[,k],d) (anisotropy.py:73-143).  With ``lazy=True`` both return handles instead
(``muygpys_amd.lazy``), which is what ``MuyGPS.make_*_tensors`` asks for.
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from muygpys_amd import lazy as _lazy
from muygpys_amd.gp.hyperparameter import NamedParam, NamedVectorParam, ScalarParam, VectorParam

from .metric import MetricFn


class DeformationFn:
    def __init__(self, metric: MetricFn, length_scale):
        self.metric = metric
        self.length_scale = length_scale

    def __call__(self, dists, **kwargs):
        raise NotImplementedError("__call__ is not implemented for base DeformationFn")

    def get_opt_params(self) -> Tuple[List[str], List[float], List[Tuple[float, float]]]:
        names: List[str] = []
        params: List[float] = []
        bounds: List[Tuple[float, float]] = []
        self.length_scale.append_lists(names, params, bounds)
        return names, params, bounds

    def _metric_name(self) -> str:
        name = getattr(self.metric, "name", None)
        if name not in ("l2", "F2"):
            raise ValueError("lazy tensors need one of the stock metrics l2 / F2")
        return name


class Isotropy(DeformationFn):
    def __init__(self, metric: MetricFn, length_scale: ScalarParam):
        if not isinstance(length_scale, ScalarParam):
            raise ValueError(f"Expected ScalarParam type for length_scale, not {type(length_scale)}")
        self.length_scale = NamedParam("length_scale", length_scale)
        self.metric = metric

    def __call__(self, dists, length_scale: Optional[float] = None, **kwargs):
        """isotropy.py:60-89: scale distances by the (possibly trial) length scale."""
        if length_scale is None:
            length_scale = self.length_scale(**kwargs)
        if isinstance(dists, _lazy.LazyDiffs):
            return dists.with_length_scale(float(length_scale))
        return self.metric.apply_length_scale(dists, length_scale)

    def pairwise_tensor(self, data, nn_indices, lazy: bool = False, **kwargs):
        """isotropy.py:92-118: (b, k, k) distances."""
        if lazy:
            return _lazy.LazyDiffs("pairwise", self._metric_name(), True, data, nn_indices)
        return self.metric.pairwise_distances(data, nn_indices)

    def crosswise_tensor(self, data, nn_data, data_indices, nn_indices, lazy: bool = False, **kwargs):
        """isotropy.py:121-161: (b, k) distances."""
        if lazy:
            return _lazy.LazyDiffs("crosswise", self._metric_name(), True, nn_data, nn_indices, data, data_indices)
        return self.metric.crosswise_distances(data, nn_data, data_indices, nn_indices)


class DifferenceIsotropy(Isotropy):
    """One length scale, applied to feature-wise DIFFERENCES before the metric: ``metric(diffs / l)``
    (isotropy.py:165-260; what the reference's experimental kernels build on).  For the stock metrics it
    is the same function of the points as :class:`Isotropy` -- ``l2(diffs / l) = l2(diffs) / l``,
    ``F2(diffs / l) = F2(diffs) / l^2`` -- so on lazy handles it ends in the same fused launch; what
    differs is the tensor the deformation hands out: differences ``(b, k[, k], d)``."""

    def __call__(self, dists, length_scale: Optional[float] = None, **kwargs):
        if length_scale is None:
            length_scale = self.length_scale(**kwargs)
        if isinstance(dists, _lazy.LazyDiffs):
            return dists.with_length_scale(float(length_scale)).reduce(self._metric_name())
        return self.metric(dists / length_scale)

    def pairwise_tensor(self, data, nn_indices, lazy: bool = False, **kwargs):
        """isotropy.py:214-237: (b, k, k, d) differences."""
        if lazy:
            return _lazy.LazyDiffs("pairwise", None, False, data, nn_indices)
        return self.metric.pairwise_differences(data, nn_indices)

    def crosswise_tensor(self, data, nn_data, data_indices, nn_indices, lazy: bool = False, **kwargs):
        """isotropy.py:240-276: (b, k, d) differences."""
        if lazy:
            return _lazy.LazyDiffs("crosswise", None, False, nn_data, nn_indices, data, data_indices)
        return self.metric.crosswise_differences(data, nn_data, data_indices, nn_indices)


class Anisotropy(DeformationFn):
    def __init__(self, metric: MetricFn, length_scale: VectorParam):
        if not isinstance(length_scale, VectorParam):
            raise ValueError(f"Expected VectorParam type for length_scale, not {type(length_scale)}")
        self.metric = metric
        self.length_scale = NamedVectorParam("length_scale", length_scale)

    def __call__(self, dists, **length_scales):
        """anisotropy.py:43-70: metric(diffs / l_vec); the last dimension must match."""
        if dists.shape[-1] != len(self.length_scale):
            raise ValueError(
                f"Difference tensor of shape {tuple(dists.shape)} must have final dimension size of "
                f"{len(self.length_scale)}"
            )
        ls = self.length_scale(**length_scales)
        if isinstance(dists, _lazy.LazyDiffs):
            return dists.with_length_scale(np.asarray(ls, dtype=np.float64)).reduce(self._metric_name())
        import torch

        if isinstance(dists, torch.Tensor) and dists.is_cuda and getattr(self.metric, "name", None) in ("l2", "F2"):
            # divide + reduce in one pass over the difference tensor
            from muygpys_amd._src.gp.tensors import hip as T

            lsv = torch.as_tensor(np.asarray(ls, dtype=np.float64), device=dists.device, dtype=dists.dtype)
            return T._reduce(dists, {"l2": 0, "F2": 1}[self.metric.name], lsv)
        if isinstance(dists, torch.Tensor):
            ls = torch.as_tensor(np.asarray(ls, dtype=np.float64), device=dists.device, dtype=dists.dtype)
        return self.metric(dists / ls)

    def pairwise_tensor(self, data, nn_indices, lazy: bool = False, **kwargs):
        """anisotropy.py:73-100: (b, k, k, d) differences."""
        if lazy:
            return _lazy.LazyDiffs("pairwise", None, False, data, nn_indices)
        return self.metric.pairwise_differences(data, nn_indices)

    def crosswise_tensor(self, data, nn_data, data_indices, nn_indices, lazy: bool = False, **kwargs):
        """anisotropy.py:103-143: (b, k, d) differences."""
        if lazy:
            return _lazy.LazyDiffs("crosswise", None, False, nn_data, nn_indices, data, data_indices)
        return self.metric.crosswise_differences(data, nn_data, data_indices, nn_indices)
