from .scalar import NamedParameter, Parameter
from .scalar import NamedParameter as NamedParam
from .scalar import Parameter as ScalarParam
from .scale import AnalyticScale, FixedScale, ScaleFn
from .tensor import TensorParam
from .vector import NamedVectorParameter, VectorParameter
from .vector import NamedVectorParameter as NamedVectorParam
from .vector import VectorParameter as VectorParam

__all__ = [
    "AnalyticScale", "FixedScale", "NamedParam", "NamedParameter", "NamedVectorParam", "NamedVectorParameter",
    "Parameter", "ScalarParam", "ScaleFn", "TensorParam", "VectorParam", "VectorParameter",
]
