"""hip implementation of the solve family (reference: src/MuyGPyS/_src/gp/muygps/numpy.py).

On materialised tensors these go through ``mgp_solve_*``: one LDS-resident Cholesky of the
(already perturbed) ``Kin`` per neighbourhood instead of the reference's LU ``linalg.solve``.  Given
the lazy handles of ``muygpys_amd.lazy`` (a complete Kin / Kcross / responses triple) the posterior
mean and variance come from ONE fused launch (``muygpys_amd.lazy_eval``), shared by the sibling calls
of one evaluation.
"""

from __future__ import annotations

import torch

from muygpys_amd import _lib, lazy, lazy_eval


def _solve(Kin, Kcross, Y, kout=1.0, want=("mean",)):
    _lib.require_cuda(Kin, Kcross, Y)
    if Kin.ndim != 3 or Kin.shape[1] != Kin.shape[2]:
        raise ValueError(f"Kin must have shape (batch, nn, nn); got {tuple(Kin.shape)}")
    K = Kin.contiguous()
    b, k, _ = K.shape
    dev, dt = K.device, K.dtype
    Kc = None if Kcross is None else Kcross.to(dt).reshape(b, k).contiguous()
    Yc = None if Y is None else Y.to(dt).reshape(b, k, -1).contiguous()
    R = 0 if Yc is None else Yc.shape[2]
    mean = torch.empty((b, R), device=dev, dtype=dt) if "mean" in want else None
    var = torch.empty((b,), device=dev, dtype=dt) if "var" in want else None
    yk = torch.empty((b, R), device=dev, dtype=dt) if "ykinvy" in want else None
    co = torch.empty((b, k, R), device=dev, dtype=dt) if "coeffs" in want else None
    info = torch.zeros(1, device=dev, dtype=torch.int32)
    rc = _lib.fn("solve", dt)(
        _lib.ptr(K), _lib.ptr(Kc), _lib.ptr(Yc), b, k, R, float(kout), _lib.ptr(mean), _lib.ptr(var),
        _lib.ptr(yk), _lib.ptr(co), _lib.ptr(info), _lib.stream_ptr(),
    )
    _lib.check(rc, "mgp_solve")
    _lib.raise_if_not_spd(info, "mgp_solve")
    return mean, var, yk, co


def _muygps_posterior_mean(Kin, Kcross, nn_targets, **kwargs):
    """numpy.py:17-41: Kcross^T Kin^-1 Y -> (b,) for (b,k) targets, (b,R) for (b,k,R)."""
    if lazy.fused_triple(Kin, Kcross, nn_targets):
        out = lazy_eval.fused(Kin, Kcross, nn_targets)
        if out is not None:
            return out[0]
    Kin, Kcross, nn_targets = lazy.force(Kin), lazy.force(Kcross), lazy.force(nn_targets)
    mean, _, _, _ = _solve(Kin, Kcross, nn_targets, want=("mean",))
    b = Kin.shape[0]
    return mean.reshape((b,) + tuple(nn_targets.shape[2:]))


def _muygps_diagonal_variance(Kin, Kcross, Kout, batch_size: int = 1, **kwargs):
    """numpy.py:44-67: Kout - Kcross^T Kin^-1 Kcross -> (b,)."""
    if isinstance(Kin, lazy.LazyCov) and isinstance(Kcross, lazy.LazyCov):
        var = lazy_eval.variance(Kin, Kcross)
        if var is not None:
            return lazy_eval.rescale_kout(var, Kout)
    Kin, Kcross = lazy.force(Kin), lazy.force(Kcross)
    kout = float(Kout) if not isinstance(Kout, torch.Tensor) else float(Kout.reshape(-1)[0].item())
    _, var, _, _ = _solve(Kin, Kcross, None, kout=kout, want=("var",))
    return var


def _muygps_fast_posterior_mean_precompute(Kin, train_nn_targets_fast, **kwargs):
    """numpy.py:88-95: coefficients Kin^-1 Y, squeezed."""
    Kin, train_nn_targets_fast = lazy.force(Kin), lazy.force(train_nn_targets_fast)
    Y = train_nn_targets_fast if train_nn_targets_fast.ndim == 3 else train_nn_targets_fast[:, :, None]
    _, _, _, co = _solve(Kin, None, Y, want=("coeffs",))
    return torch.squeeze(co)


def _fast_mean_fused(Kcross, coeffs_tensor):
    """A lazy crosswise covariance handle and the gathered coefficients ``coeffs[closest_index]`` (b, k[, R]) --
    what the reference's workflow hands over (examples/from_indices.py:113-118) -- through the fused prediction kernel
    (``mgp_fast_posterior_mean_*``: gather, distances, kernel, dot product in one launch; nothing of size (b, k) is
    formed).  None when the handle is not a plain crosswise covariance the kernel evaluates, or the neighbourhood has
    more slots than a wavefront (``k + 1 > 64``: mgp_fast_mean.hip serves one test point per 32 or 64 lanes) -- the
    caller then materialises the handle and takes the reference's einsum."""
    from muygpys_amd.fused import FusedUnsupported, KernelSpec, fast_posterior_mean

    c = Kcross.diffs
    if not (c.kind == "crosswise" and c.reduced and c.metric in ("l2", "F2") and Kcross.kernel != "matern_gen"):
        return None
    if not (isinstance(coeffs_tensor, torch.Tensor) and coeffs_tensor.is_cuda and coeffs_tensor.ndim in (2, 3)):
        return None
    b, k = c.nn_indices.shape
    if tuple(coeffs_tensor.shape[:2]) != (b, k) or coeffs_tensor.dtype != c.dtype or k + 1 > 64:
        return None
    spec = KernelSpec(Kcross.kernel, c.metric, 1.0 if c.length_scale is None else c.length_scale, 0.0)
    rows = torch.arange(b, device=coeffs_tensor.device)
    try:
        out = fast_posterior_mean(spec, c.data, c.nn_data, c.data_indices, c.nn_indices, coeffs_tensor, rows)
    except FusedUnsupported:
        return None
    return torch.squeeze(out)


def _muygps_fast_posterior_mean(Kcross, coeffs_tensor, **kwargs):
    """numpy.py:70-77: einsum('ij,ijk->ik')."""
    if isinstance(Kcross, lazy.LazyCov):
        out = _fast_mean_fused(Kcross, lazy.force(coeffs_tensor))
        if out is not None:
            return out
    Kcross, coeffs_tensor = lazy.force(Kcross), lazy.force(coeffs_tensor)
    _lib.require_cuda(Kcross, coeffs_tensor)
    C = torch.atleast_3d(coeffs_tensor)  # (k,) -> (1,k,1), (b,k) -> (b,k,1), like np.atleast_3d
    return torch.squeeze(torch.einsum("ij,ijk->ik", Kcross, C))


def _mmuygps_fast_posterior_mean(Kcross, coeffs_tensor, **kwargs):
    """numpy.py:80-85."""
    _lib.require_cuda(Kcross, coeffs_tensor)
    return torch.einsum("ijk,ijk->ik", Kcross, coeffs_tensor)
