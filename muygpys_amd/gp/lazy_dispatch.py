"""Default ``_backend_*`` callables of the functor layer under the hip backend.

These are the hip family functions themselves (``muygpys_amd._src``): each accepts either
materialised device tensors (then it runs the per-function kernel) or the lazy handles of
``muygpys_amd.lazy`` (then a complete triple runs the fused kernel once and the siblings reuse its
cached outputs, ``muygpys_amd.lazy_eval``).  The names below are the ones the functor classes take
as ``_backend_*`` keyword defaults (gp/muygps.py:98-101, gp/kernels/matern.py:61-68, rbf.py:77,
gp/noise/homoscedastic.py:46, gp/hyperparameter/scale.py:165).
"""

from __future__ import annotations

from muygpys_amd._src.gp.kernels import hip as _K
from muygpys_amd._src.gp.muygps import hip as _M
from muygpys_amd._src.gp.noise import hip as _N
from muygpys_amd._src.optimize.scale import hip as _S

rbf_fn = _K._rbf_fn
matern_05_fn = _K._matern_05_fn
matern_15_fn = _K._matern_15_fn
matern_25_fn = _K._matern_25_fn
matern_inf_fn = _K._matern_inf_fn
matern_gen_fn = _K._matern_gen_fn

homoscedastic_perturb = _N._homoscedastic_perturb
heteroscedastic_perturb = _N._heteroscedastic_perturb

posterior_mean = _M._muygps_posterior_mean
diagonal_variance = _M._muygps_diagonal_variance
fast_posterior_mean = _M._muygps_fast_posterior_mean
fast_posterior_mean_precompute = _M._muygps_fast_posterior_mean_precompute

analytic_scale_optim = _S._analytic_scale_optim
analytic_scale_optim_unnormalized = _S._analytic_scale_optim_unnormalized
