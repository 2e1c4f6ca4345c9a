"""Deep-kernel layers over the differentiable HIP path (reference package: MuyGPyS/torch)."""

from .muygps_layer import MultivariateMuyGPs_layer, MuyGPs_layer  # noqa: F401
