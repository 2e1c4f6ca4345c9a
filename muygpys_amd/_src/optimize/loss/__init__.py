"""Loss family (reference name list: _src/optimize/loss/__init__.py:8-23)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _mse_fn
    _cross_entropy_fn
    _lool_fn
    _lool_fn_unscaled
    _pseudo_huber_fn
    _looph_fn
    """,
)
