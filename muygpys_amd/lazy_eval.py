"""Evaluation of lazy handles: one fused launch per (Kin, Kcross, responses) triple.

The hip family functions (``muygpys_amd._src``) call in here when they are handed the handles of
``muygpys_amd.lazy`` instead of tensors.  ``fused`` runs ``muygpys_amd.fused.posterior_mean_var``
once and caches ``(mean, var [Kout = 1], ykinvy)`` on the cache dict the differently decorated copies
of one ``Kin`` share, so that the sibling calls of one objective evaluation -- posterior mean,
diagonal variance, analytic scale (optimize/loss.py:158-176) -- cost one launch together.

A cache entry holds the objects it was computed from and is valid only for exactly those objects at
exactly those versions (``is`` + ``Tensor._version``): a new Kcross, an in-place edit of the
responses or of a noise tensor miss the cache instead of returning a stale launch.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import lazy


def _noise_key(noise):
    if noise is None:
        return ("scalar", 0.0)
    if isinstance(noise, torch.Tensor) and noise.ndim >= 1:
        return ("tensor", noise)
    return ("scalar", float(noise))


def _version(x) -> int:
    return x._version if isinstance(x, torch.Tensor) else 0


class _Entry:
    __slots__ = ("noise", "noise_v", "cross", "targets", "targets_v", "differentiable", "value")

    def __init__(self, noise, cross, targets, differentiable, value):
        self.noise, self.noise_v = noise, _version(noise[1])
        self.cross, self.targets, self.targets_v = cross, targets, _version(targets)
        self.differentiable, self.value = differentiable, value

    def noise_matches(self, noise) -> bool:
        if self.noise[0] != noise[0]:
            return False
        if noise[0] == "scalar":
            return self.noise[1] == noise[1]
        return self.noise[1] is noise[1] and self.noise_v == _version(noise[1])

    def targets_match(self, targets) -> bool:
        return self.targets is targets and self.targets_v == _version(targets)


def _entries(Kin: lazy.LazyCov):
    return Kin.cache.setdefault("entries", [])


def _targets_tensor(nn_targets):
    return nn_targets.targets if isinstance(nn_targets, lazy.LazyTargets) else nn_targets


def _wants_grad(*xs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


def _spec(Kin: lazy.LazyCov):
    from .fused import KernelSpec

    a = Kin.diffs
    return KernelSpec(
        kernel=Kin.kernel, metric=a.metric, length_scale=1.0 if a.length_scale is None else a.length_scale,
        noise=0.0 if Kin.noise is None else Kin.noise, smoothness=Kin.smoothness,
    )


def fused(Kin: lazy.LazyCov, Kcross: lazy.LazyCov, nn_targets, differentiable: Optional[bool] = None):
    """(mean, var_unscaled_with_Kout_1, ykinvy, info) of a lazy triple, computed once per
    (noise, Kcross, responses) and cached on the shared Kin cache -- or None when no fused kernel serves
    the model (:class:`muygpys_amd.fused.FusedUnsupported`: the general-smoothness Matern on fp64 tables
    or in an oversized small batch); the caller then materialises.

    ``nn_targets``: a :class:`lazy.LazyTargets` handle (responses gathered inside the kernel) or the
    already gathered (b, k[, R]) tensor (``mgp_posterior_gathered_*``).  When a feature table, the
    responses, or a tensor-valued length scale / noise requires grad (deep-kernel training through
    MuyGPs_layer, torch/muygps_layer.py:129-164) the launch goes through :mod:`muygpys_amd.autograd`,
    whose backward is the HIP vector-Jacobian kernel; that entry carries no ``ykinvy`` (the scale is a
    constant of the layer, as in the reference)."""
    from . import _lib
    from .fused import FusedUnsupported, posterior_mean_var

    a, c = Kin.diffs, Kcross.diffs
    if Kin.cache.get("unsupported"):
        return None
    tg = _targets_tensor(nn_targets)
    gathered = not isinstance(nn_targets, lazy.LazyTargets)
    if differentiable is None:
        differentiable = (not gathered) and _wants_grad(a.nn_data, c.data, tg, a.length_scale, Kin.noise)
    nk = _noise_key(Kin.noise)
    for e in _entries(Kin):
        if e.differentiable == bool(differentiable) and e.noise_matches(nk) and e.cross is c and e.targets_match(tg):
            return e.value
    spec = _spec(Kin)
    if differentiable and Kin.kernel == "matern_gen":
        return None  # (the backward kernels know the closed-form kernels only)
    if differentiable:
        from .autograd import posterior

        value = posterior(spec, c.data, a.nn_data, c.data_indices, a.nn_indices, tg) + (None, None)
        # the differentiable entry lives next to the plain ones (the scale launch must not evict it)
        _entries(Kin).append(_Entry(nk, c, tg, True, value))
        return value
    info = torch.zeros(1, dtype=torch.int32, device=a.device)
    try:
        value = posterior_mean_var(
            spec, c.data, a.nn_data, c.data_indices, a.nn_indices, tg, want_ykinvy=True, info=info, gathered=gathered,
        ) + (info,)
    except FusedUnsupported:
        Kin.cache["unsupported"] = True  # (shared by the decorated copies of this Kin: asked once)
        return None
    _lib.raise_if_not_spd(info, "fused posterior")
    # one evaluation at a time: the hyper-parameters changed -> older plain entries are dead
    Kin.cache["entries"] = [e for e in _entries(Kin) if e.differentiable]
    _entries(Kin).append(_Entry(nk, c, tg, False, value))
    return value


def variance(Kin: lazy.LazyCov, Kcross: lazy.LazyCov):
    """Unscaled variance for Kout = 1 of a lazy (Kin, Kcross) pair, or None when the pair does not
    describe one fused launch.  The variance does not depend on the responses: any cached launch of
    this Kin / Kcross serves."""
    nk = _noise_key(Kin.noise)
    for e in _entries(Kin):
        if e.noise_matches(nk) and e.cross is Kcross.diffs:
            return e.value[1]
    a = Kin.diffs
    dummy = lazy.LazyTargets(torch.zeros((a.nn_data.shape[0],), device=Kin.device, dtype=Kin.dtype), a.nn_indices)
    if lazy.fused_triple(Kin, Kcross, dummy):
        out = fused(Kin, Kcross, dummy, differentiable=False)
        return None if out is None else out[1]
    return None


def rescale_kout(var_kout1: torch.Tensor, Kout):
    kout = float(Kout) if not isinstance(Kout, torch.Tensor) else float(Kout.reshape(-1)[0].item())
    return var_kout1 if kout == 1.0 else var_kout1 + (kout - 1.0)


def analytic_scale(Kin: lazy.LazyCov, nn_targets):
    """sigma^2 = sum_b y^T K^-1 y / (b k) of a lazy Kin and its responses (optimize/scale/numpy.py:
    18-34), from the ``ykinvy`` output of the evaluation's fused launch; None when the handles do not
    allow it (the caller then materialises)."""
    from . import _lib
    from . import distributed as _D

    a = Kin.diffs
    if not (a.kind == "pairwise" and a.reduced and a.metric in ("l2", "F2")):
        return None
    tg = _targets_tensor(nn_targets)
    b, k = a.nn_indices.shape
    if isinstance(nn_targets, lazy.LazyTargets):
        if tg.ndim > 1 and tg.shape[1] != 1:
            raise ValueError(f"cannot reshape array of size {b * k * tg.shape[1]} into shape ({b},{k},1)")
    elif not (isinstance(tg, torch.Tensor) and tuple(tg.shape[:2]) == (b, k)):
        return None
    elif tg.numel() != b * k:
        raise ValueError(f"cannot reshape array of size {tg.numel()} into shape ({b},{k},1)")
    nk = _noise_key(Kin.noise)
    yk = None
    for e in _entries(Kin):
        if e.noise_matches(nk) and e.targets_match(tg) and e.value[2] is not None:
            yk = e.value[2]
            break
    if yk is None:
        # no sibling launch yet (MuyGPS.optimize_scale has no crosswise tensor): run the fused kernel
        # with neighbour 0 standing in as the query -- its mean / variance outputs are meaningless and
        # dropped, y^T K^-1 y does not depend on the query at all
        stand_in = lazy.LazyCov(
            lazy.LazyDiffs("crosswise", a.metric, True, a.nn_data, a.nn_indices, a.nn_data,
                           a.nn_indices[:, 0].contiguous(), a.length_scale),
            Kin.kernel, smoothness=Kin.smoothness,
        )
        out = fused(Kin, stand_in, nn_targets, differentiable=False)
        if out is None:
            return None
        yk = out[2]
    from .config import config

    out = _lib.column_sums(yk.reshape(b, -1).contiguous())
    if _D.reductions_active():  # sharded batch: global sum / global count (scale/mpi.py:16-37)
        tot = torch.cat([out.sum().reshape(1), torch.tensor([float(b)], device=out.device, dtype=torch.float64)])
        _D.reduce_if_sharded_(tot)
        val = (tot[0] / (tot[1] * k)).to(yk.dtype)
    else:
        val = (out.sum() / (b * k)).to(yk.dtype)
    # under integration.install() the value goes to the reference's AnalyticScale._set, which calls
    # len() on it (gp/hyperparameter/scale.py:47-52): a 0-d tensor refuses that, a (1,) tensor passes
    return val.reshape(1) if config.state.lazy_tensors else val
