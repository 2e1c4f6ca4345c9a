from muygpys_amd._src.util import _collect_implementation

(
    _muygps_posterior_mean,
    _muygps_diagonal_variance,
    _muygps_fast_posterior_mean,
    _muygps_fast_posterior_mean_precompute,
    _mmuygps_fast_posterior_mean,
) = _collect_implementation(
    "muygpys_amd._src.gp.muygps",
    "_muygps_posterior_mean",
    "_muygps_diagonal_variance",
    "_muygps_fast_posterior_mean",
    "_muygps_fast_posterior_mean_precompute",
    "_mmuygps_fast_posterior_mean",
)
