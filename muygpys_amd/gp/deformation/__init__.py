from .deformation_fn import Anisotropy, DeformationFn, DifferenceIsotropy, Isotropy
from .metric import F2, MetricFn, l2

__all__ = ["Anisotropy", "DeformationFn", "DifferenceIsotropy", "F2", "Isotropy", "MetricFn", "l2"]
