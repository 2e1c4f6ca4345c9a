"""hip implementation of the noise family (reference: src/MuyGPyS/_src/gp/noise/numpy.py)."""

from __future__ import annotations

import torch

from muygpys_amd import _lib, lazy


def _homoscedastic_perturb(Kin, noise_variance):
    """numpy.py:9-27 (3-D case; the 5-D block form belongs to the out-of-scope shear kernel)."""
    if isinstance(Kin, lazy.LazyCov):
        return Kin.perturbed(float(noise_variance))  # the nugget is added while the fused kernel assembles K
    _lib.require_cuda(Kin)
    if Kin.ndim != 3:
        raise ValueError(
            f"homoscedastic perturbation is not implemented for tensors of shape {tuple(Kin.shape)}"
        )
    x = Kin.contiguous()
    b, k, _ = x.shape
    out = torch.empty_like(x)
    rc = _lib.fn("perturb", x.dtype)(
        _lib.ptr(x), b, k, _lib.NOISE_SCALAR, float(noise_variance), None, _lib.ptr(out), _lib.stream_ptr()
    )
    _lib.check(rc, "mgp_perturb")
    return out


def _heteroscedastic_perturb(Kin, noise_variances):
    """numpy.py:56-67: Kin[b,i,i] += eps[b,i]."""
    noise_variances = lazy.force(noise_variances)
    if isinstance(Kin, lazy.LazyCov):
        return Kin.perturbed(noise_variances)
    _lib.require_cuda(Kin, noise_variances)
    x = Kin.contiguous()
    b, k, _ = x.shape
    nz = noise_variances.to(dtype=x.dtype).reshape(b, k).contiguous()
    out = torch.empty_like(x)
    rc = _lib.fn("perturb", x.dtype)(
        _lib.ptr(x), b, k, _lib.NOISE_BATCH, 0.0, _lib.ptr(nz), _lib.ptr(out), _lib.stream_ptr()
    )
    _lib.check(rc, "mgp_perturb")
    return out


def _shear_perturb33(Kin, noise_variance):
    """numpy.py:30-53 belongs to the experimental shear kernel, outside the hot path."""
    raise NotImplementedError("The hip backend does not implement the experimental shear noise model.")
