"""Kernel functors ``Matern`` and ``RBF`` (reference contract:
src/MuyGPyS/gp/kernels/{kernel_fn,matern,rbf}.py).

A kernel functor composes ``kernel_fn(deformation(diffs, **length_scale kwargs), **other)``
and exposes that closure to the optimiser through ``get_opt_fn`` with every non-fixed
hyper-parameter as a keyword argument.  Matern picks a closed-form backend function only
when the smoothness is FIXED at 0.5 / 1.5 / 2.5 / inf (matern.py:61-81); anything else is
the general Bessel form (``mgp_matern_gen_*``: K_nu evaluated on the device), which runs on
materialised distances instead of the fused launch.
"""

from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

from muygpys_amd._src.util import auto_str
from muygpys_amd.gp import lazy_dispatch as _ld
from muygpys_amd.gp.deformation import DeformationFn, F2, Isotropy, l2
from muygpys_amd.gp.hyperparameter import NamedParam, ScalarParam


class KernelFn:
    def __init__(self, deformation: DeformationFn):
        self._hyperparameters: Dict = dict()
        self.deformation = deformation
        self._make_base()

    def _make_base(self):
        self.deformation.length_scale.populate(self._hyperparameters)

    def _make(self):
        raise NotImplementedError("_make is not implemented for base KernelFn")

    def set_params(self, **kwargs) -> None:
        for name in kwargs:
            self._hyperparameters[name]._set(kwargs[name])

    def __call__(self, diffs, **kwargs):
        raise NotImplementedError("__call__ is not implemented for base KernelFn")

    def get_opt_fn(self) -> Callable:
        raise NotImplementedError("get_opt_fn is not implemented for base KernelFn")

    def Kout(self):
        raise NotImplementedError("Kout is not implemented for base KernelFn")

    def get_opt_params(self) -> Tuple[List[str], List[float], List[Tuple[float, float]]]:
        names: List[str] = []
        params: List[float] = []
        bounds: List[Tuple[float, float]] = []
        self.deformation.length_scale.append_lists(names, params, bounds)
        return names, params, bounds

    def __str__(self) -> str:
        return "\n".join(f"{n} : {p()} - {p.get_bounds()}" for n, p in self._hyperparameters.items())


def _set_matern_fn(
    smoothness: ScalarParam,
    _backend_05_fn: Callable = _ld.matern_05_fn,
    _backend_15_fn: Callable = _ld.matern_15_fn,
    _backend_25_fn: Callable = _ld.matern_25_fn,
    _backend_inf_fn: Callable = _ld.matern_inf_fn,
    _backend_gen_fn: Callable = _ld.matern_gen_fn,
):
    if smoothness.fixed():
        nu = smoothness()
        if nu == 0.5:
            return _backend_05_fn
        if nu == 1.5:
            return _backend_15_fn
        if nu == 2.5:
            return _backend_25_fn
        if nu == math.inf:
            return _backend_inf_fn
    return _backend_gen_fn


@auto_str
class Matern(KernelFn):
    def __init__(
        self,
        smoothness: ScalarParam = ScalarParam(0.5),
        deformation: DeformationFn = Isotropy(l2, length_scale=ScalarParam(1.0)),
        _backend_ones: Optional[Callable] = None,
        _backend_zeros: Optional[Callable] = None,
        _backend_squeeze: Optional[Callable] = None,
        **_backend_fns,
    ):
        super().__init__(deformation=deformation)
        self.smoothness = NamedParam("smoothness", smoothness)
        self._backend_ones = _backend_ones
        self._backend_zeros = _backend_zeros
        self._backend_squeeze = _backend_squeeze
        self._backend_fns = _backend_fns
        self._make()

    def _make(self):
        super()._make_base()
        self.smoothness.populate(self._hyperparameters)
        self._kernel_fn = _set_matern_fn(self.smoothness, **self._backend_fns)
        self._predef_fn = self.smoothness.apply_fn(self._kernel_fn)
        self._fn = self.deformation.length_scale.apply_embedding_fn(self._predef_fn, self.deformation)

    def __call__(self, diffs, **kwargs):
        return self._fn(diffs, **kwargs)

    def Kout(self, **kwargs):
        """Prior variance at the query: the scalar 1 (matern.py:170-171)."""
        if self._backend_ones is not None and self._backend_squeeze is not None:
            return self._backend_squeeze(self._backend_ones((1, 1)))
        return 1.0

    def get_opt_params(self):
        names, params, bounds = super().get_opt_params()
        self.smoothness.append_lists(names, params, bounds)
        return names, params, bounds

    def get_opt_fn(self) -> Callable:
        return self._fn


@auto_str
class RBF(KernelFn):
    def __init__(
        self,
        deformation: DeformationFn = Isotropy(F2, length_scale=ScalarParam(1.0)),
        _backend_fn: Callable = _ld.rbf_fn,
        _backend_ones: Optional[Callable] = None,
        _backend_zeros: Optional[Callable] = None,
        _backend_squeeze: Optional[Callable] = None,
    ):
        super().__init__(deformation=deformation)
        self._backend_ones = _backend_ones
        self._backend_zeros = _backend_zeros
        self._backend_squeeze = _backend_squeeze
        self._kernel_fn = _backend_fn
        self._make()

    def _make(self):
        super()._make_base()
        self._fn = self.deformation.length_scale.apply_embedding_fn(self._kernel_fn, self.deformation)

    def __call__(self, diffs, **kwargs):
        return self._fn(diffs, **kwargs)

    def Kout(self, **kwargs):
        if self._backend_ones is not None and self._backend_squeeze is not None:
            return self._backend_squeeze(self._backend_ones((1, 1)))
        return 1.0

    def get_opt_fn(self) -> Callable:
        return self._fn
