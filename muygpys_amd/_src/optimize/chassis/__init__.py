"""Outer optimiser drivers (reference name list: _src/optimize/chassis/__init__.py:8-15)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _scipy_optimize
    _bayes_opt_optimize
    """,
)
