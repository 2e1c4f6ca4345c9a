"""Kernel family: elementwise covariance functions (reference name list: _src/gp/kernels/__init__.py:8-23)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _rbf_fn
    _matern_05_fn
    _matern_15_fn
    _matern_25_fn
    _matern_inf_fn
    _matern_gen_fn
    """,
)
