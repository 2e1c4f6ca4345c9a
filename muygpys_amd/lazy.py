"""Lazy tensor handles: how the reference's call sequence reaches the fused HIP kernel.

A MuyGPyS caller writes (gp/muygps.py:406-551, gp/kernels/matern.py:148-168, :164-259)

    crosswise, pairwise, nn_targets = muygps.make_predict_tensors(...)
    Kin, Kcross = muygps.kernel(pairwise), muygps.kernel(crosswise)
    mean = muygps.posterior_mean(Kin, Kcross, nn_targets)
    var  = muygps.posterior_variance(Kin, Kcross)

and the reference materialises a (b,k,k[,d]) tensor at every line.  Under the hip backend
``make_*_tensors`` return the light handles below instead; the kernel functor and the noise
model decorate them; and ``posterior_mean`` / ``posterior_variance`` / the analytic scale
recognise a complete (Kin, Kcross, targets) triple and issue ONE fused launch
(``muygpys_amd.fused``), whose three outputs are cached on the handle so the second and
third call are free.  Every handle can be forced with ``.materialize()`` (or by passing it
to any backend function), which runs the per-function kernels -- so code that really wants
the tensors still gets them, computed on the GPU.
"""

from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Any, Dict, Optional, Tuple

import torch


@dataclass
class LazyDiffs:
    """What ``deformation.pairwise_tensor`` / ``crosswise_tensor`` stand for.

    kind "pairwise": data (n, d) + nn_indices (b, k)  ->  (b, k, k[, d])
    kind "crosswise": data (n_q, d), data_indices (b,), nn_data, nn_indices -> (b, k[, d])
    reduced: the metric has been applied (Isotropy hands distances to the kernel,
    isotropy.py:92-161); False = raw differences (Anisotropy, anisotropy.py:73-143).
    """

    kind: str
    metric: str
    reduced: bool
    nn_data: torch.Tensor
    nn_indices: torch.Tensor
    data: Optional[torch.Tensor] = None
    data_indices: Optional[torch.Tensor] = None
    length_scale: Any = None  # set by the deformation functor: float or (d,) sequence

    @property
    def shape(self) -> Tuple[int, ...]:
        b, k = self.nn_indices.shape
        base = (b, k, k) if self.kind == "pairwise" else (b, k)
        if self.reduced:
            return base
        d = 1 if self.nn_data.ndim == 1 else self.nn_data.shape[1]
        return base + (d,)

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def dtype(self):
        return self.nn_data.dtype

    @property
    def device(self):
        return self.nn_data.device

    def with_length_scale(self, length_scale) -> "LazyDiffs":
        return replace(self, length_scale=length_scale)

    def materialize(self) -> torch.Tensor:
        from muygpys_amd._src.gp.tensors import hip as T

        if self.kind == "pairwise":
            if self.reduced:
                return T._pairwise_distances(self.nn_data, self.nn_indices, self.metric)
            return T._pairwise_tensor(self.nn_data, self.nn_indices)
        if self.reduced:
            return T._crosswise_distances(self.data, self.nn_data, self.data_indices, self.nn_indices, self.metric)
        return T._crosswise_tensor(self.data, self.nn_data, self.data_indices, self.nn_indices)


@dataclass
class LazyCov:
    """What ``kernel(LazyDiffs)`` stands for: Kin (pairwise) or Kcross (crosswise)."""

    diffs: LazyDiffs
    kernel: str
    noise: Any = None          # attached by the noise model's perturb(): float or tensor
    cache: Dict = field(default_factory=dict, repr=False, compare=False)

    @property
    def shape(self):
        b, k = self.diffs.nn_indices.shape
        return (b, k, k) if self.diffs.kind == "pairwise" else (b, k)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def dtype(self):
        return self.diffs.dtype

    @property
    def device(self):
        return self.diffs.device

    def perturbed(self, noise) -> "LazyCov":
        # the cache dict is shared on purpose: mean, variance and scale of one evaluation
        # see differently decorated copies of the same Kin
        return LazyCov(self.diffs, self.kernel, noise, self.cache)

    def materialize(self) -> torch.Tensor:
        """kernel(metric(diffs / length_scale)) [+ nugget] through the per-function kernels."""
        from muygpys_amd._src.gp.kernels import hip as K
        from muygpys_amd._src.gp.noise import hip as N
        from muygpys_amd._src.gp.tensors import hip as T

        d = self.diffs
        ls = d.length_scale
        if d.reduced:
            scale = 1.0 / float(ls) if d.metric == "l2" else 1.0 / float(ls) ** 2
            out = K._apply(d.materialize(), self.kernel, scale)
        else:
            lsv = torch.as_tensor(ls, device=d.device, dtype=d.dtype).reshape(-1)
            dist = T._reduce(d.materialize(), {"l2": 0, "F2": 1}[d.metric], lsv)
            out = K._apply(dist, self.kernel, 1.0)
        if self.noise is not None and d.kind == "pairwise":
            if isinstance(self.noise, torch.Tensor) and self.noise.ndim >= 1:
                out = N._heteroscedastic_perturb(out, self.noise)
            else:
                out = N._homoscedastic_perturb(out, float(self.noise))
        return out


@dataclass
class LazyTargets:
    """What ``train_targets[batch_nn_indices]`` stands for."""

    targets: torch.Tensor
    nn_indices: torch.Tensor

    @property
    def shape(self):
        return tuple(self.nn_indices.shape) + tuple(self.targets.shape[1:])

    @property
    def ndim(self):
        return len(self.shape)

    def materialize(self) -> torch.Tensor:
        return self.targets[self.nn_indices]


def is_lazy(x) -> bool:
    return isinstance(x, (LazyDiffs, LazyCov, LazyTargets))


def force(x):
    """Materialise a handle; pass anything else through."""
    return x.materialize() if is_lazy(x) else x


def fused_triple(Kin, Kcross, nn_targets) -> bool:
    """True when (Kin, Kcross, targets) describe one fused launch: same neighbour table and
    index tensor, same kernel and deformation."""
    if not (isinstance(Kin, LazyCov) and isinstance(Kcross, LazyCov) and isinstance(nn_targets, LazyTargets)):
        return False
    a, c = Kin.diffs, Kcross.diffs
    return (
        a.kind == "pairwise" and c.kind == "crosswise" and Kin.kernel == Kcross.kernel
        and a.nn_indices is c.nn_indices and a.nn_indices is nn_targets.nn_indices
        and a.nn_data is c.nn_data and a.metric == c.metric and a.reduced == c.reduced
        and _same_ls(a.length_scale, c.length_scale)
    )


def _same_ls(x, y) -> bool:
    import numpy as np

    return np.array_equal(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
