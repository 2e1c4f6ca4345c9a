from muygpys_amd._src.util import _collect_implementation

(
    _scipy_optimize,
    _bayes_opt_optimize,
) = _collect_implementation(
    "muygpys_amd._src.optimize.chassis",
    "_scipy_optimize",
    "_bayes_opt_optimize",
)
