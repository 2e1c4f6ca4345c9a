"""Tensor-valued, never-optimised hyper-parameter (reference: gp/hyperparameter/tensor.py);
only the heteroscedastic noise model uses it."""

from __future__ import annotations


class TensorParam:
    def __init__(self, val):
        self._set_val(val)

    def _set(self, val=None) -> None:
        if val is not None:
            self._set_val(val)

    def _set_val(self, val) -> None:
        if isinstance(val, str):
            raise ValueError("TensorParam class does not support strings.")
        if not hasattr(val, "shape"):
            raise ValueError(f"Non-array tensor hyperparameter type {type(val)} is not allowed.")
        self._val = val

    def __call__(self):
        return self._val

    def fixed(self) -> bool:
        return True

    def get_bounds(self):
        return "fixed"
