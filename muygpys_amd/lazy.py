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


def _force_tree(x):
    if isinstance(x, (list, tuple)):
        return type(x)(_force_tree(v) for v in x)
    if isinstance(x, dict):
        return {k: _force_tree(v) for k, v in x.items()}
    return force(x)


class _Lazy:
    """Anything a handle does not know how to do lazily is done on the materialised tensor: torch
    functions (``__torch_function__``), tensor methods and attributes, indexing and arithmetic."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*_force_tree(args), **_force_tree(kwargs or {}))

    def __getattr__(self, name):
        if name.startswith("__") or name in ("cache",):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, item):
        return self.materialize()[item]

    def __len__(self):
        return self.shape[0]

    def __mul__(self, other):
        return self.materialize() * force(other)

    __rmul__ = __mul__

    def __add__(self, other):
        return self.materialize() + force(other)

    __radd__ = __add__

    def __sub__(self, other):
        return self.materialize() - force(other)

    def __rsub__(self, other):
        return force(other) - self.materialize()

    def __neg__(self):
        return -self.materialize()

    def __pow__(self, p):
        return self.materialize() ** p


def _is_scalar_like(x) -> bool:
    if isinstance(x, (int, float)):
        return True
    if isinstance(x, torch.Tensor):
        return x.numel() == 1
    try:
        import numpy as np

        return np.ndim(x) == 0 or np.size(x) == 1
    except Exception:
        return False


def _as_float(x) -> float:
    if isinstance(x, torch.Tensor):
        return float(x.detach().reshape(-1)[0])
    import numpy as np

    return float(np.asarray(x).reshape(-1)[0])


@dataclass(eq=False)
class LazyDiffs(_Lazy):
    """What the tensor family's ``_pairwise_tensor`` / ``_crosswise_tensor`` (and the deformation
    functors' ``pairwise_tensor`` / ``crosswise_tensor``) stand for.

    kind "pairwise": data (n, d) + nn_indices (b, k)  ->  (b, k, k[, d])
    kind "crosswise": data (n_q, d), data_indices (b,), nn_data, nn_indices -> (b, k[, d])

    The handle follows the reference's order of operations without computing anything:
      raw differences         (metric None,  reduced False)   _src/gp/tensors/numpy.py:47-69
      / per-feature scales    (length_scale vector)           Anisotropy.__call__, anisotropy.py:70
      metric: _l2 / _F2       (metric set,   reduced True)    numpy.py:89-94
      / l  or  / l^2          (length_scale scalar)           Isotropy.__call__, metric.py:241,264
    ``reduced``: the handle stands for DISTANCES (b, k[, k]); otherwise for differences (..., d).
    """

    kind: str
    metric: Optional[str]
    reduced: bool
    nn_data: torch.Tensor
    nn_indices: torch.Tensor
    data: Optional[torch.Tensor] = None
    data_indices: Optional[torch.Tensor] = None
    length_scale: Any = None  # float, or (d,) sequence / tensor

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

    def reduce(self, metric: str) -> "LazyDiffs":
        """``_l2`` / ``_F2`` of a difference handle."""
        if self.reduced:
            raise ValueError("the metric has already been applied to this tensor")
        return replace(self, metric=metric, reduced=True)

    def __truediv__(self, other):
        other = force(other)
        if not self.reduced and self.length_scale is None:
            # differences / per-feature length scales (Anisotropy) or / one length scale
            d = self.shape[-1]
            n = 1 if _is_scalar_like(other) else (other.numel() if isinstance(other, torch.Tensor) else len(other))
            if n == 1:
                return self.with_length_scale(_as_float(other))
            if n == d and (not isinstance(other, torch.Tensor) or other.ndim == 1):
                return self.with_length_scale(other)
        elif self.reduced and self.length_scale is None and _is_scalar_like(other):
            # distances / l (l2) or / l^2 (F2): metric.py:241,264
            v = _as_float(other)
            return self.with_length_scale(v if self.metric == "l2" else v**0.5)
        return self.materialize() / other

    def _raw(self) -> torch.Tensor:
        from muygpys_amd._src.gp.tensors import hip as T

        if self.kind == "pairwise":
            return T._pairwise_tensor_now(self.nn_data, self.nn_indices)
        return T._crosswise_tensor_now(self.data, self.nn_data, self.data_indices, self.nn_indices)

    def _ls_vector(self):
        ls = self.length_scale
        if ls is None or _is_scalar_like(ls):
            return None
        return torch.as_tensor(ls, device=self.device, dtype=self.dtype).reshape(-1)

    def materialize(self) -> torch.Tensor:
        """The tensor the handle stands for (differences, or -- once the metric is applied --
        distances with whatever length scale has been attached)."""
        from muygpys_amd._src.gp.tensors import hip as T

        lsv = self._ls_vector()
        if not self.reduced:
            raw = self._raw()
            if self.length_scale is None:
                return raw
            return raw / (lsv if lsv is not None else _as_float(self.length_scale))
        if lsv is not None:  # per-feature scales act before the metric
            return T._reduce(self._raw(), {"l2": 0, "F2": 1}[self.metric], lsv)
        if self.kind == "pairwise":
            out = T._pairwise_distances(self.nn_data, self.nn_indices, self.metric)
        else:
            out = T._crosswise_distances(self.data, self.nn_data, self.data_indices, self.nn_indices, self.metric)
        if self.length_scale is not None:
            ell = _as_float(self.length_scale)
            out = out * (1.0 / ell if self.metric == "l2" else 1.0 / ell**2)
        return out

    def scaled_distances(self) -> torch.Tensor:
        """metric(differences / length_scale): what the deformation hands to the kernel function."""
        if not self.reduced:
            raise ValueError("no metric has been applied to this difference tensor")
        return self.materialize()


@dataclass(eq=False)
class LazyCov(_Lazy):
    """What ``kernel(LazyDiffs)`` stands for: Kin (pairwise) or Kcross (crosswise)."""

    diffs: LazyDiffs
    kernel: str
    noise: Any = None          # attached by the noise model's perturb(): float or tensor
    cache: Dict = field(default_factory=dict, repr=False, compare=False)
    smoothness: Optional[float] = None  # kernel "matern_gen": the Matern smoothness nu

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
        return LazyCov(self.diffs, self.kernel, noise, self.cache, self.smoothness)

    def materialize(self) -> torch.Tensor:
        """kernel(metric(diffs / length_scale)) [+ nugget] through the per-function kernels."""
        from muygpys_amd._src.gp.kernels import hip as K
        from muygpys_amd._src.gp.noise import hip as N

        d = self.diffs
        if self.kernel == "matern_gen":
            out = K._matern_gen_fn(d.scaled_distances(), self.smoothness)
        else:
            out = K._apply(d.scaled_distances(), self.kernel, 1.0)
        if self.noise is not None and d.kind == "pairwise":
            if isinstance(self.noise, torch.Tensor) and self.noise.ndim >= 1:
                out = N._heteroscedastic_perturb(out, self.noise)
            else:
                out = N._homoscedastic_perturb(out, float(self.noise))
        return out


@dataclass(eq=False)
class LazyTargets(_Lazy):
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
        return torch.Tensor.__getitem__(self.targets.as_subclass(torch.Tensor), self.nn_indices)


def is_lazy(x) -> bool:
    return isinstance(x, (LazyDiffs, LazyCov, LazyTargets))


def force(x):
    """Materialise a handle; pass anything else through."""
    return x.materialize() if is_lazy(x) else x


def _same_tensor(x, y) -> bool:
    return x is y or (
        isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.data_ptr() == y.data_ptr()
        and x.shape == y.shape and x.stride() == y.stride() and x.dtype == y.dtype
    )


def fused_triple(Kin, Kcross, nn_targets) -> bool:
    """True when (Kin, Kcross, targets) describe one fused launch: same neighbour table and
    index tensor, same kernel and deformation.  ``nn_targets`` is a :class:`LazyTargets` handle or
    the already gathered (b, k[, R]) tensor (the reference's functor layer gathers it itself)."""
    if not (isinstance(Kin, LazyCov) and isinstance(Kcross, LazyCov)):
        return False
    a, c = Kin.diffs, Kcross.diffs
    b, k = a.nn_indices.shape
    if isinstance(nn_targets, LazyTargets):
        if not _same_tensor(a.nn_indices, nn_targets.nn_indices):
            return False
    elif not (isinstance(nn_targets, torch.Tensor) and tuple(nn_targets.shape[:2]) == (b, k)):
        return False
    return (
        a.kind == "pairwise" and c.kind == "crosswise" and a.reduced and c.reduced and Kin.kernel == Kcross.kernel
        and Kin.smoothness == Kcross.smoothness
        and _same_tensor(a.nn_indices, c.nn_indices) and _same_tensor(a.nn_data, c.nn_data)
        and a.metric == c.metric and a.metric in ("l2", "F2") and _same_ls(a.length_scale, c.length_scale)
    )


def _same_ls(x, y) -> bool:
    import numpy as np

    def host(v):
        if v is None:
            return np.asarray(1.0)
        if isinstance(v, torch.Tensor):
            return v.detach().double().cpu().numpy()
        return np.asarray(v, dtype=np.float64)

    return x is y or np.array_equal(host(x), host(y))
