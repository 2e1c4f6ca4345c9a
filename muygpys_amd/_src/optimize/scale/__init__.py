from muygpys_amd._src.util import _collect_implementation

(
    _analytic_scale_optim,
    _analytic_scale_optim_unnormalized,
) = _collect_implementation(
    "muygpys_amd._src.optimize.scale",
    "_analytic_scale_optim",
    "_analytic_scale_optim_unnormalized",
)
