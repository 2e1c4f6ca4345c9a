"""Scalar hyper-parameters (reference contract: src/MuyGPyS/gp/hyperparameter/scalar.py:29-348).

A ``Parameter`` is a value plus optimisation bounds (``"fixed"`` or a ``(low, high)`` pair);
the value may be given as ``"sample"`` / ``"log_sample"`` to draw it from the bounds.
``NamedParameter`` attaches the keyword under which the optimiser passes trial values and
provides the closure plumbing (``apply_fn``, ``apply_embedding_fn``, ``filter_kwargs``) that
routes those keywords to the kernel / deformation.
"""

from __future__ import annotations

from numbers import Number
from typing import Callable, Dict, List, Sequence, Tuple, Union

import numpy as np


class _Bounds:
    """Optimisation bounds of one scalar: ``"fixed"`` or a ``(low, high)`` pair of numbers.  Parsing, the
    range test of a value (with the reference's 1e-5 slack) and the two sampling rules live here; ``Parameter``
    keeps the plain attributes (``_bounds``, ``_fixed``) the rest of the package reads."""

    __slots__ = ("pair", "fixed")

    def __init__(self, spec):
        if isinstance(spec, str):
            if spec != "fixed":
                raise ValueError(f"Unknown bound option {spec}.")
            self.pair, self.fixed = (0.0, 0.0), True
            return
        if not hasattr(spec, "__iter__"):
            raise ValueError(f"Unknown bound optiom {spec} of a non-iterable type {type(spec)}.")
        if len(spec) != 2:
            raise ValueError(f"Provided hyperparameter optimization bounds have unsupported length {len(spec)}.")
        bad = next((v for v in spec if not isinstance(v, Number)), None)
        if bad is not None:
            raise ValueError(f"Nonscalar {bad} of type {type(bad)} is not a supported hyperparameter bound type.")
        lo, hi = float(spec[0]), float(spec[1])
        if lo > hi:
            raise ValueError(f"Lower bound {lo} is not lesser than upper bound {hi}.")
        self.pair, self.fixed = (lo, hi), False

    def check(self, val: float) -> None:
        if self.fixed:
            return
        lo, hi = self.pair
        if val < lo - 1e-5:
            raise ValueError(f"Hyperparameter value {val} is lesser than the optimization lower bound {lo}")
        if val > hi + 1e-5:
            raise ValueError(f"Hyperparameter value {val} is greater than the optimization upper bound {hi}")

    def draw(self, how: str) -> float:
        if self.fixed:
            raise ValueError(f"Fixed bounds do not support string value ({how}) prompts.")
        lo, hi = self.pair
        if how == "sample":
            return float(np.random.uniform(low=lo, high=hi))
        if how == "log_sample":
            return float(np.exp(np.random.uniform(low=np.log(lo), high=np.log(hi))))
        raise ValueError(f"Unsupported string hyperparameter value {how}.")


def _is_scalar_tensor(val) -> bool:
    shape = getattr(val, "shape", None)
    return shape is not None and len(shape) == 0


class Parameter:
    def __init__(self, val: Union[str, float], bounds: Union[str, Tuple[float, float]] = "fixed"):
        self._set_bounds(bounds)
        self._set_val(val)

    def __str__(self, **kwargs):
        return f"{type(self).__name__}({self._val}, {'fixed' if self._fixed else self._bounds})"

    def _set(self, rhs: "Parameter") -> None:
        self._val, self._bounds, self._fixed = rhs._val, rhs._bounds, rhs._fixed

    def _set_bounds(self, bounds) -> None:
        parsed = _Bounds(bounds)
        self._bounds, self._fixed = parsed.pair, parsed.fixed

    def _limits(self) -> _Bounds:
        return _Bounds("fixed" if self._fixed else self._bounds)

    def _set_val(self, val) -> None:
        if isinstance(val, str):
            val = self._limits().draw(val)
            # under sharded reductions (the mpi-backend layout: distributed.enable_sharded_mode() process-wide, like
            # the reference's MUYGPYS_BACKEND=mpi, or inside a sharded_reductions block) every rank starts from
            # rank 0's draw (scalar.py:145-146: bcast(root=0) when _is_mpi_mode()).  A plain torch.distributed job
            # that samples on some ranks only is left alone (a broadcast there would hang); a model built that way
            # is still safe to optimise sharded: the drivers start every rank from rank 0's values
            # (_src/optimize/chassis/hip.py: _get_opt_lists)
            from muygpys_amd import distributed as _D

            if _D.reductions_active():
                val = _D.broadcast_scalar(val, _D.active_group())
        if isinstance(val, Sequence) or (hasattr(val, "__len__") and not _is_scalar_tensor(val)):
            raise ValueError(f"Nonscalar hyperparameter value {val} is not allowed.")
        val = float(val)
        self._limits().check(val)
        self._val = val

    def __call__(self, **kwargs) -> float:
        return self._val

    def get_bounds(self) -> Tuple[float, float]:
        return self._bounds

    def fixed(self) -> bool:
        return self._fixed


class _KeywordRouting:
    """Closure plumbing of the named parameters (scalar.py:314-334, vector.py:102-121): a subclass says which
    keywords are its own (``filter_kwargs``: (mine-with-defaults, rest)); wrapping a kernel function or a
    kernel-of-deformation is then the same for a scalar and for a vector."""

    def filter_kwargs(self, **kwargs) -> Tuple[Dict, Dict]:  # pragma: no cover - provided by the subclasses
        raise NotImplementedError

    def apply_fn(self, fn: Callable) -> Callable:
        def applied_fn(*args, **kwargs):
            mine, rest = self.filter_kwargs(**kwargs)
            return fn(*args, **mine, **rest)

        return applied_fn

    def apply_embedding_fn(self, fn: Callable, deformation_fn: Callable) -> Callable:
        """kernel(deformation(dists, <own keywords>), **other_hyper)."""

        def embedded_fn(dists, *args, **kwargs):
            mine, rest = self.filter_kwargs(**kwargs)
            return fn(deformation_fn(dists, **mine), *args, **rest)

        return embedded_fn


class NamedParameter(_KeywordRouting, Parameter):
    def __init__(self, name: str, param: Parameter):
        self._set(param)
        self._name = name

    def name(self) -> str:
        return self._name

    def filter_kwargs(self, **kwargs) -> Tuple[Dict, Dict]:
        """scalar.py:321-325: split off this parameter's keyword (defaulting to the stored value)."""
        rest = dict(kwargs)
        return {self._name: rest.pop(self._name, self())}, rest

    def append_lists(self, names: List[str], params: List[float], bounds: List[Tuple[float, float]]):
        if not self.fixed():
            names.append(self._name)
            params.append(self())
            bounds.append(self.get_bounds())

    def populate(self, hyperparameters: Dict) -> None:
        hyperparameters[self._name] = self
