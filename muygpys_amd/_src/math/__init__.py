from muygpys_amd._src.util import _collect_implementation

_NAMES = (
    "all", "allclose", "arange", "argmax", "array", "atleast_1d", "atleast_2d", "assign", "corrcoef", "cov",
    "cholesky", "eye", "exp", "diagonal", "divide", "iarray", "inf", "int32", "int64", "itype", "isclose",
    "farray", "float32", "float64", "ftype", "full", "linalg", "linspace", "log", "logical_or", "max", "mean",
    "median", "min", "ndarray", "ones", "outer", "parameter", "prod", "repeat", "reshape", "sqrt", "squeeze",
    "sum", "tile", "unique", "vstack", "where", "zeros",
)
(
    all, allclose, arange, argmax, array, atleast_1d, atleast_2d, assign, corrcoef, cov, cholesky, eye, exp,
    diagonal, divide, iarray, inf, int32, int64, itype, isclose, farray, float32, float64, ftype, full, linalg,
    linspace, log, logical_or, max, mean, median, min, ndarray, ones, outer, parameter, prod, repeat, reshape,
    sqrt, squeeze, sum, tile, unique, vstack, where, zeros,
) = _collect_implementation("muygpys_amd._src.math", *_NAMES)


def promote(x):
    return x if isinstance(x, ndarray) else array(x)
