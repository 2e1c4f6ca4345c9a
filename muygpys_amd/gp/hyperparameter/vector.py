"""Vector hyper-parameters: one scalar ``Parameter`` per feature dimension
(reference contract: src/MuyGPyS/gp/hyperparameter/vector.py:19-143).  The named form
exposes its elements to the optimiser as ``<name>0 ... <name>{d-1}``."""

from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import numpy as np

from .scalar import NamedParameter, Parameter, _KeywordRouting


class VectorParameter:
    def __init__(self, *params: Parameter):
        self._params = list(params)
        self._named = False

    def __len__(self) -> int:
        return len(self._params)

    def set_name(self, name):
        self._name = name
        self._named = True

    def __str__(self, **kwargs):
        return f"{type(self).__name__}(" + ", ".join(str(p) for p in self._params) + ")"

    def __call__(self, **kwargs):
        return np.array([p() for p in self._params], dtype=np.float64)

    def fixed(self) -> bool:
        return all(p.fixed() for p in self._params)


class NamedVectorParameter(_KeywordRouting, VectorParameter):
    def __init__(self, name: str, param: VectorParameter):
        self._params = [NamedParameter(name + str(i), p) for i, p in enumerate(param._params)]
        self._name = name

    def name(self) -> str:
        return self._name

    def set_defaults(self, **params) -> Dict:
        for p in self._params:
            params.setdefault(p.name(), p())
        return params

    def filter_kwargs(self, **kwargs) -> Tuple[Dict, Dict]:
        """vector.py:92-100: keywords that start with the name are this vector's elements."""
        mine = {key: val for key, val in kwargs.items() if key.startswith(self._name)}
        rest = {key: val for key, val in kwargs.items() if not key.startswith(self._name)}
        return self.set_defaults(**mine), rest

    def __call__(self, **kwargs):
        mine, _ = self.filter_kwargs(**kwargs)
        # element order = index order, whatever order the optimiser passed the keywords in
        return np.array([float(mine[p.name()]) for p in self._params], dtype=np.float64)

    def append_lists(self, names: List[str], params: List[float], bounds: List[Tuple[float, float]]):
        for p in self._params:
            p.append_lists(names, params, bounds)

    def populate(self, hyperparameters: Dict) -> None:
        for p in self._params:
            p.populate(hyperparameters)
