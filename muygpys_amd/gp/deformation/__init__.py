from .deformation_fn import Anisotropy, DeformationFn, Isotropy
from .metric import F2, MetricFn, l2

__all__ = ["Anisotropy", "DeformationFn", "F2", "Isotropy", "MetricFn", "l2"]
