"""Metric functors ``l2`` / ``F2`` (reference contract: src/MuyGPyS/gp/deformation/metric.py).

A ``MetricFn`` bundles the backend functions that reduce difference tensors and the rule
that applies a length scale (l2: x / l, F2: x / l**2, metric.py:241,264).  The constructor
keywords are the reference's (including ``pairwise_diffferences_fn`` with three f's -- the
typo is part of its API).  Two optional extras name the fused distance kernels, which form
distances straight from the feature table without the (b, k, k, d) intermediate.
"""

from __future__ import annotations

from typing import Callable, Optional

from muygpys_amd._src.gp.tensors import _crosswise_tensor, _F2, _l2, _pairwise_tensor
from muygpys_amd._src.gp.tensors import hip as _T


class MetricFn:
    def __init__(
        self,
        differences_metric_fn: Callable,
        crosswise_differences_fn: Callable,
        pairwise_diffferences_fn: Callable,
        apply_length_scale_fn: Callable,
        name: Optional[str] = None,
        crosswise_distances_fn: Optional[Callable] = None,
        pairwise_distances_fn: Optional[Callable] = None,
    ):
        self._differences_metric_fn = differences_metric_fn
        self._crosswise_differences_fn = crosswise_differences_fn
        self._pairwise_differences_fn = pairwise_diffferences_fn
        self._apply_length_scale_fn = apply_length_scale_fn
        self.name = name
        self._crosswise_distances_fn = crosswise_distances_fn
        self._pairwise_distances_fn = pairwise_distances_fn

    def __call__(self, *args, **kwargs):
        return self._differences_metric_fn(*args, **kwargs)

    def crosswise_differences(self, data, nn_data, data_indices, nn_indices, **kwargs):
        """metric.py:71-111: (b, k, d)."""
        return self._crosswise_differences_fn(data, nn_data, data_indices, nn_indices)

    def crosswise_distances(self, data, nn_data, data_indices, nn_indices, **kwargs):
        """metric.py:113-155: (b, k)."""
        if self._crosswise_distances_fn is not None:
            return self._crosswise_distances_fn(data, nn_data, data_indices, nn_indices)
        return self(self.crosswise_differences(data, nn_data, data_indices, nn_indices))

    def pairwise_differences(self, data, nn_indices, **kwargs):
        """metric.py:157-184: (b, k, k, d)."""
        return self._pairwise_differences_fn(data, nn_indices)

    def pairwise_distances(self, data, nn_indices, **kwargs):
        """metric.py:186-214: (b, k, k)."""
        if self._pairwise_distances_fn is not None:
            return self._pairwise_distances_fn(data, nn_indices)
        return self(self.pairwise_differences(data, nn_indices))

    def apply_length_scale(self, dists, length_scale):
        """metric.py:216-234."""
        return self._apply_length_scale_fn(dists, length_scale)


l2 = MetricFn(
    differences_metric_fn=_l2,
    crosswise_differences_fn=_crosswise_tensor,
    pairwise_diffferences_fn=_pairwise_tensor,
    apply_length_scale_fn=lambda x, y: x / y,
    name="l2",
    crosswise_distances_fn=lambda *a: _T._crosswise_distances(*a, "l2"),
    pairwise_distances_fn=lambda *a: _T._pairwise_distances(*a, "l2"),
)
F2 = MetricFn(
    differences_metric_fn=_F2,
    crosswise_differences_fn=_crosswise_tensor,
    pairwise_diffferences_fn=_pairwise_tensor,
    apply_length_scale_fn=lambda x, y: x / y**2,
    name="F2",
    crosswise_distances_fn=lambda *a: _T._crosswise_distances(*a, "F2"),
    pairwise_distances_fn=lambda *a: _T._pairwise_distances(*a, "F2"),
)
