"""Noise models (reference contract: src/MuyGPyS/gp/noise/{homoscedastic,heteroscedastic,null}.py).

``perturb_fn(fn)`` wraps a function of ``Kin`` so that it sees ``Kin + nugget``; the
homoscedastic wrapper also accepts a trial ``noise=`` keyword, used while the nugget is being
optimised (homoscedastic.py:90-115).
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple, Union

from muygpys_amd import lazy as _lazy
from muygpys_amd.gp import lazy_dispatch as _ld
from muygpys_amd.gp.hyperparameter import NamedParam, ScalarParam, TensorParam


class NoiseFn:
    def __call__(self):
        pass

    def fixed(self) -> bool:
        return True

    def perturb(self, Kin, **kwargs):
        return Kin

    def perturb_fn(self, fn: Callable) -> Callable:
        return fn

    def append_lists(self, names, params, bounds):
        pass


class HomoscedasticNoise(NamedParam, NoiseFn):
    def __init__(
        self,
        val: Union[str, float],
        bounds: Union[str, Tuple[float, float]] = "fixed",
        _backend_fn: Callable = _ld.homoscedastic_perturb,
    ):
        super().__init__("noise", ScalarParam(val, bounds))
        if not self.fixed() and (self._bounds[0] < 0.0 or self._bounds[1] < 0.0):
            raise ValueError(f"Homoscedastic noise optimization bounds {self._bounds} are not strictly positive!")
        self._perturb_fn = _backend_fn

    def perturb(self, Kin, noise: Optional[float] = None, **kwargs):
        if noise is None:
            noise = self._val
        return self._perturb_fn(Kin, noise)

    def perturb_fn(self, fn: Callable) -> Callable:
        def perturbed_fn(Kin, *args, noise=None, **kwargs):
            return fn(self.perturb(Kin, noise=noise), *args, **kwargs)

        return perturbed_fn

    def append_lists(self, names, params, bounds):
        NamedParam.append_lists(self, names, params, bounds)


class HeteroscedasticNoise(TensorParam, NoiseFn):
    """Per-observation nugget: a (batch, nn) tensor aligned with the batch's neighbourhoods
    (build it with ``make_heteroscedastic_tensor``), never optimised."""

    def __init__(self, val, _backend_fn: Callable = _ld.heteroscedastic_perturb):
        super().__init__(val)
        v = _lazy.force(val)
        if bool((v.flatten() < 0).sum() > 0):
            raise ValueError("Heteroscedastic noise values are not strictly non-negative!")
        self._perturb_fn = _backend_fn

    def perturb(self, Kin, **kwargs):
        return self._perturb_fn(Kin, self._val)

    def perturb_fn(self, fn: Callable) -> Callable:
        def perturbed_fn(Kin, *args, **kwargs):
            return fn(self.perturb(Kin), *args, **kwargs)

        return perturbed_fn

    def fixed(self) -> bool:
        return True


class NullNoise(ScalarParam, NoiseFn):
    def __init__(self, *args, **kwargs):
        self.val = 0.0
        self.bounds = "fixed"

    def __call__(self, *args, **kwargs):
        return 0.0

    def fixed(self) -> bool:
        return True

    def perturb(self, Kin, **kwargs):
        return Kin

    def perturb_fn(self, fn: Callable) -> Callable:
        return fn


__all__ = ["HeteroscedasticNoise", "HomoscedasticNoise", "NoiseFn", "NullNoise"]
