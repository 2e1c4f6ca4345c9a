from muygpys_amd._src.util import _collect_implementation

(
    _homoscedastic_perturb,
    _heteroscedastic_perturb,
    _shear_perturb33,
) = _collect_implementation(
    "muygpys_amd._src.gp.noise",
    "_homoscedastic_perturb",
    "_heteroscedastic_perturb",
    "_shear_perturb33",
)
