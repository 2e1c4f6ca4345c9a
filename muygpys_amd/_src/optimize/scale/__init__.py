"""Analytic sigma^2 family (reference name list: _src/optimize/scale/__init__.py:8-15)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _analytic_scale_optim
    _analytic_scale_optim_unnormalized
    """,
)
