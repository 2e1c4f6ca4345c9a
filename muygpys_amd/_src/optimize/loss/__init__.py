from muygpys_amd._src.util import _collect_implementation

(
    _mse_fn,
    _cross_entropy_fn,
    _lool_fn,
    _lool_fn_unscaled,
    _pseudo_huber_fn,
    _looph_fn,
) = _collect_implementation(
    "muygpys_amd._src.optimize.loss",
    "_mse_fn",
    "_cross_entropy_fn",
    "_lool_fn",
    "_lool_fn_unscaled",
    "_pseudo_huber_fn",
    "_looph_fn",
)
