from muygpys_amd._src.util import _collect_implementation

(
    _rbf_fn,
    _matern_05_fn,
    _matern_15_fn,
    _matern_25_fn,
    _matern_inf_fn,
    _matern_gen_fn,
) = _collect_implementation(
    "muygpys_amd._src.gp.kernels",
    "_rbf_fn",
    "_matern_05_fn",
    "_matern_15_fn",
    "_matern_25_fn",
    "_matern_inf_fn",
    "_matern_gen_fn",
)
