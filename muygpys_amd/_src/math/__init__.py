"""The ``mm`` array facade (reference name list: _src/math/__init__.py:8-109)."""

from muygpys_amd._src.util import export_backend

_NAMES = tuple(export_backend(
    __name__,
    globals(),
    """
    all allclose arange argmax array atleast_1d atleast_2d assign corrcoef cov cholesky eye exp
    diagonal divide iarray inf int32 int64 itype isclose farray float32 float64 ftype full linalg
    linspace log logical_or max mean median min ndarray ones outer parameter prod repeat reshape
    sqrt squeeze sum tile unique vstack where zeros
    """,
))


def promote(x):
    return x if isinstance(x, ndarray) else array(x)  # noqa: F821 (bound above)
