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


class Parameter:
    def __init__(self, val: Union[str, float], bounds: Union[str, Tuple[float, float]] = "fixed"):
        self._set_bounds(bounds)
        self._set_val(val)

    def __str__(self, **kwargs):
        return f"{type(self).__name__}({self._val}, {'fixed' if self._fixed else self._bounds})"

    def _set(self, rhs: "Parameter") -> None:
        self._val, self._bounds, self._fixed = rhs._val, rhs._bounds, rhs._fixed

    def _sample_val(self, val: str) -> float:
        if self._fixed:
            raise ValueError(f"Fixed bounds do not support string value ({val}) prompts.")
        lo, hi = self._bounds
        if val == "sample":
            new = float(np.random.uniform(low=lo, high=hi))
        elif val == "log_sample":
            new = float(np.exp(np.random.uniform(low=np.log(lo), high=np.log(hi))))
        else:
            raise ValueError(f"Unsupported string hyperparameter value {val}.")
        # under sharded reductions (the mpi-backend layout) every rank starts from rank 0's draw
        # (scalar.py:145-146: bcast(root=0) when _is_mpi_mode()); a plain torch.distributed job that
        # samples on some ranks only is left alone
        from muygpys_amd import distributed as _D

        return _D.broadcast_scalar(new, _D.active_group()) if _D.reductions_active() else new

    def _set_val(self, val) -> None:
        if isinstance(val, str):
            val = self._sample_val(val)
        if isinstance(val, Sequence) or (hasattr(val, "__len__") and not _is_scalar_tensor(val)):
            raise ValueError(f"Nonscalar hyperparameter value {val} is not allowed.")
        val = float(val)
        if not self._fixed:
            if val < self._bounds[0] - 1e-5:
                raise ValueError(
                    f"Hyperparameter value {val} is lesser than the optimization lower bound {self._bounds[0]}"
                )
            if val > self._bounds[1] + 1e-5:
                raise ValueError(
                    f"Hyperparameter value {val} is greater than the optimization upper bound {self._bounds[1]}"
                )
        self._val = val

    def _set_bounds(self, bounds) -> None:
        if isinstance(bounds, str):
            if bounds != "fixed":
                raise ValueError(f"Unknown bound option {bounds}.")
            self._bounds, self._fixed = (0.0, 0.0), True
            return
        if not hasattr(bounds, "__iter__"):
            raise ValueError(f"Unknown bound optiom {bounds} of a non-iterable type {type(bounds)}.")
        if len(bounds) != 2:
            raise ValueError(
                f"Provided hyperparameter optimization bounds have unsupported length {len(bounds)}."
            )
        for v in bounds:
            if not isinstance(v, Number):
                raise ValueError(f"Nonscalar {v} of type {type(v)} is not a supported hyperparameter bound type.")
        lo, hi = float(bounds[0]), float(bounds[1])
        if lo > hi:
            raise ValueError(f"Lower bound {lo} is not lesser than upper bound {hi}.")
        self._bounds, self._fixed = (lo, hi), False

    def __call__(self, **kwargs) -> float:
        return self._val

    def get_bounds(self) -> Tuple[float, float]:
        return self._bounds

    def fixed(self) -> bool:
        return self._fixed


def _is_scalar_tensor(val) -> bool:
    shape = getattr(val, "shape", None)
    return shape is not None and len(shape) == 0


class NamedParameter(Parameter):
    def __init__(self, name: str, param: Parameter):
        self._set(param)
        self._name = name

    def name(self) -> str:
        return self._name

    def apply_fn(self, fn: Callable) -> Callable:
        """scalar.py:314-319: default the keyword to the stored value."""

        def applied_fn(*args, **kwargs):
            kwargs.setdefault(self._name, self())
            return fn(*args, **kwargs)

        return applied_fn

    def filter_kwargs(self, **kwargs) -> Tuple[Dict, Dict]:
        """scalar.py:321-325: split off this parameter's keyword."""
        mine = {key: val for key, val in kwargs.items() if key == self._name}
        rest = {key: val for key, val in kwargs.items() if key != self._name}
        mine.setdefault(self._name, self())
        return mine, rest

    def apply_embedding_fn(self, fn: Callable, deformation_fn: Callable) -> Callable:
        """scalar.py:327-334: kernel(deformation(dists, length_scale=...), **other_hyper)."""

        def embedded_fn(dists, *args, **kwargs):
            mine, rest = self.filter_kwargs(**kwargs)
            return fn(deformation_fn(dists, **mine), *args, **rest)

        return embedded_fn

    def append_lists(self, names: List[str], params: List[float], bounds: List[Tuple[float, float]]):
        if not self.fixed():
            names.append(self._name)
            params.append(self())
            bounds.append(self.get_bounds())

    def populate(self, hyperparameters: Dict) -> None:
        hyperparameters[self._name] = self
