"""Default ``_backend_*`` callables of the functor layer under the hip backend.

Each one accepts either materialised device tensors (then it is exactly the hip backend
function of ``muygpys_amd._src``) or the lazy handles of ``muygpys_amd.lazy`` (then a
complete triple runs the fused kernel once and the siblings reuse its cached outputs).
"""

from __future__ import annotations

import torch

from muygpys_amd import lazy
from muygpys_amd._src.gp.kernels import hip as _K
from muygpys_amd._src.gp.muygps import hip as _M
from muygpys_amd._src.gp.noise import hip as _N
from muygpys_amd._src.optimize.scale import hip as _S

_KERNEL_OF_FN = {
    "_rbf_fn": "rbf", "_matern_05_fn": "matern05", "_matern_15_fn": "matern15",
    "_matern_25_fn": "matern25", "_matern_inf_fn": "maternInf",
}


def _lazy_kernel(name: str, hip_fn):
    def kernel_fn(dists, **kwargs):
        if isinstance(dists, lazy.LazyDiffs):
            return lazy.LazyCov(dists, name)
        return hip_fn(dists, **kwargs)

    kernel_fn.__name__ = hip_fn.__name__
    kernel_fn.__doc__ = hip_fn.__doc__
    return kernel_fn


rbf_fn = _lazy_kernel("rbf", _K._rbf_fn)
matern_05_fn = _lazy_kernel("matern05", _K._matern_05_fn)
matern_15_fn = _lazy_kernel("matern15", _K._matern_15_fn)
matern_25_fn = _lazy_kernel("matern25", _K._matern_25_fn)
matern_inf_fn = _lazy_kernel("maternInf", _K._matern_inf_fn)


def _scaled_distances(d: lazy.LazyDiffs) -> torch.Tensor:
    """metric(diffs / length_scale) of a lazy difference handle, materialised (what the deformation
    functor would have handed to the kernel function: isotropy.py:60-89, anisotropy.py:43-70)."""
    from muygpys_amd._src.gp.tensors import hip as T

    ls = d.length_scale
    if d.reduced:
        scale = 1.0 / float(ls) if d.metric == "l2" else 1.0 / float(ls) ** 2
        return d.materialize() * scale
    lsv = torch.as_tensor(ls, device=d.device, dtype=d.dtype).reshape(-1)
    return T._reduce(d.materialize(), {"l2": 0, "F2": 1}[d.metric], lsv)


def matern_gen_fn(dists, smoothness, **kwargs):
    """General smoothness (a free or non-special nu): not one of the fused kernels' closed forms, so
    the distances are materialised, the Bessel-function kernel (``mgp_matern_gen_*``) applied, and
    the posterior goes through ``mgp_solve_*`` on the materialised tensors."""
    if isinstance(dists, lazy.LazyDiffs):
        dists = _scaled_distances(dists)
    return _K._matern_gen_fn(dists, smoothness, **kwargs)


def homoscedastic_perturb(Kin, noise_variance):
    if isinstance(Kin, lazy.LazyCov):
        return Kin.perturbed(float(noise_variance))
    return _N._homoscedastic_perturb(Kin, noise_variance)


def heteroscedastic_perturb(Kin, noise_variances):
    if isinstance(Kin, lazy.LazyCov):
        return Kin.perturbed(lazy.force(noise_variances))
    return _N._heteroscedastic_perturb(Kin, lazy.force(noise_variances))


def _noise_key(noise):
    if noise is None:
        return ("scalar", 0.0)
    if isinstance(noise, torch.Tensor) and noise.ndim >= 1:
        return ("tensor", noise.data_ptr(), tuple(noise.shape))
    return ("scalar", float(noise))


def _wants_grad(*xs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


def _fused(Kin: lazy.LazyCov, Kcross: lazy.LazyCov, nn_targets: lazy.LazyTargets, differentiable=None):
    """(mean, var_unscaled_with_Kout_1, ykinvy) of a lazy triple, computed once per
    (noise, Kcross, targets) and cached on the shared Kin cache.

    When a feature table, the targets, or a tensor-valued length scale / noise requires grad
    (deep-kernel training through MuyGPs_layer, torch/muygps_layer.py:129-164) the launch goes
    through :mod:`muygpys_amd.autograd`, whose backward is the HIP vector-Jacobian kernel; that
    entry carries no ``ykinvy`` (the scale is a constant of the layer, as in the reference)."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    a, c = Kin.diffs, Kcross.diffs
    if differentiable is None:
        differentiable = _wants_grad(a.nn_data, c.data, nn_targets.targets, a.length_scale, Kin.noise)
    key = (_noise_key(Kin.noise), id(Kcross.diffs), id(nn_targets.targets), bool(differentiable))
    hit = Kin.cache.get(key)
    if hit is None and differentiable:
        from muygpys_amd.autograd import posterior

        spec = KernelSpec(
            kernel=Kin.kernel, metric=a.metric, length_scale=a.length_scale,
            noise=0.0 if Kin.noise is None else Kin.noise,
        )
        hit = posterior(spec, c.data, a.nn_data, c.data_indices, a.nn_indices, nn_targets.targets) + (None, None)
        Kin.cache[key] = hit
    if hit is None:
        spec = KernelSpec(
            kernel=Kin.kernel, metric=a.metric, length_scale=a.length_scale,
            noise=0.0 if Kin.noise is None else Kin.noise,
        )
        info = torch.zeros(1, dtype=torch.int32, device=a.device)
        hit = posterior_mean_var(
            spec, c.data, a.nn_data, c.data_indices, a.nn_indices, nn_targets.targets, want_ykinvy=True, info=info
        ) + (info,)
        from muygpys_amd import _lib

        _lib.raise_if_not_spd(info, "fused posterior")
        Kin.cache.clear()  # one evaluation at a time: hyper-parameters changed -> old entries are dead
        Kin.cache[key] = hit
    return hit


def posterior_mean(Kin, Kcross, nn_targets, **kwargs):
    if lazy.fused_triple(Kin, Kcross, nn_targets):
        return _fused(Kin, Kcross, nn_targets)[0]
    return _M._muygps_posterior_mean(lazy.force(Kin), lazy.force(Kcross), lazy.force(nn_targets), **kwargs)


def diagonal_variance(Kin, Kcross, Kout, batch_size: int = 1, **kwargs):
    if isinstance(Kin, lazy.LazyCov) and isinstance(Kcross, lazy.LazyCov):
        # the variance does not depend on the responses: reuse any cached launch of this Kin/Kcross
        nk = _noise_key(Kin.noise)
        for key, hit in Kin.cache.items():
            if key[0] == nk and key[1] == id(Kcross.diffs):
                return _rescale_kout(hit[1], Kout)
        dummy = lazy.LazyTargets(_zero_targets(Kin), Kin.diffs.nn_indices)
        if lazy.fused_triple(Kin, Kcross, dummy):
            return _rescale_kout(_fused(Kin, Kcross, dummy)[1], Kout)
    return _M._muygps_diagonal_variance(lazy.force(Kin), lazy.force(Kcross), Kout, batch_size=batch_size, **kwargs)


def _zero_targets(Kin: lazy.LazyCov):
    n = Kin.diffs.nn_data.shape[0]
    return torch.zeros((n,), device=Kin.device, dtype=Kin.dtype)


def _rescale_kout(var_kout1, Kout):
    kout = float(Kout) if not isinstance(Kout, torch.Tensor) else float(Kout.reshape(-1)[0].item())
    return var_kout1 if kout == 1.0 else var_kout1 + (kout - 1.0)


def analytic_scale_optim(Kin, nn_targets, batch_dim_count: int = 1, **kwargs):
    if isinstance(Kin, lazy.LazyCov) and isinstance(nn_targets, lazy.LazyTargets):
        if nn_targets.targets.ndim > 1 and nn_targets.targets.shape[1] != 1:
            b, k = nn_targets.nn_indices.shape
            raise ValueError(f"cannot reshape array of size {b * k * nn_targets.targets.shape[1]} into shape ({b},{k},1)")
        nk = _noise_key(Kin.noise)
        for key, hit in Kin.cache.items():
            if key[0] == nk and key[2] == id(nn_targets.targets) and hit[2] is not None:
                return _scale_from_ykinvy(hit[2], Kin)
        # no sibling launch yet (MuyGPS.optimize_scale has no crosswise tensor): run the fused
        # kernel with neighbour 0 standing in as the query -- its mean/variance outputs are
        # meaningless and dropped, y^T K^-1 y does not depend on the query at all
        a = Kin.diffs
        if a.kind == "pairwise":
            stand_in = lazy.LazyCov(
                lazy.LazyDiffs("crosswise", a.metric, a.reduced, a.nn_data, a.nn_indices, a.nn_data,
                               a.nn_indices[:, 0].contiguous(), a.length_scale),
                Kin.kernel,
            )
            return _scale_from_ykinvy(_fused(Kin, stand_in, nn_targets, differentiable=False)[2], Kin)
    return _S._analytic_scale_optim(lazy.force(Kin), lazy.force(nn_targets), batch_dim_count=batch_dim_count, **kwargs)


def analytic_scale_optim_unnormalized(Kin, nn_targets, **kwargs):
    return _S._analytic_scale_optim_unnormalized(lazy.force(Kin), lazy.force(nn_targets), **kwargs)


def _scale_from_ykinvy(yk: torch.Tensor, Kin: lazy.LazyCov):
    from muygpys_amd import _lib

    b, k = Kin.diffs.nn_indices.shape
    out = _lib.column_sums(yk.reshape(b, -1).contiguous())
    from muygpys_amd import distributed as _D

    if _D.reductions_active():  # sharded batch: global sum / global count (scale/mpi.py:16-37)
        tot = torch.cat([out.sum().reshape(1), torch.tensor([float(b)], device=out.device, dtype=torch.float64)])
        _D.reduce_if_sharded_(tot)
        return (tot[0] / (tot[1] * k)).to(yk.dtype)
    return (out.sum() / (b * k)).to(yk.dtype)


def fast_posterior_mean(Kcross, coeffs_tensor, **kwargs):
    return _M._muygps_fast_posterior_mean(lazy.force(Kcross), lazy.force(coeffs_tensor), **kwargs)


def fast_posterior_mean_precompute(Kin, train_nn_targets_fast, **kwargs):
    return _M._muygps_fast_posterior_mean_precompute(lazy.force(Kin), lazy.force(train_nn_targets_fast), **kwargs)
