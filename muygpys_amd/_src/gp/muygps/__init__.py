"""Solve family: posterior mean / variance and the fast-mean helpers (reference name list: _src/gp/muygps/__init__.py:8-21)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _muygps_posterior_mean
    _muygps_diagonal_variance
    _muygps_fast_posterior_mean
    _muygps_fast_posterior_mean_precompute
    _mmuygps_fast_posterior_mean
    """,
)
