"""Noise family: nugget perturbations (reference name list: _src/gp/noise/__init__.py:8-15)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _homoscedastic_perturb
    _heteroscedastic_perturb
    _shear_perturb33
    """,
)
