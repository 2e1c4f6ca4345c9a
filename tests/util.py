import numpy as np

# north_star tolerances vs the numpy fp64 reference, applied as
# |a-b| <= rtol*|b| + rtol*rms(b)  
RTOL = {"float64": 1e-5, "float32": 1e-3}


def assert_close(got, ref, rtol, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    rms = float(np.sqrt(np.mean(ref**2))) if ref.size else 0.0
    err = np.abs(got - ref)
    bound = rtol * np.abs(ref) + rtol * rms
    bad = ~(err <= bound)
    assert not bad.any(), (
        f"{what}: {bad.sum()} of {ref.size} outside tol {rtol}; max abs err {np.nanmax(err):.3e}, rms(ref) {rms:.3e}"
    )


def assert_rel_close(got, ref, rtol, what="", floor=0.0):
    """Relative bound |a-b| <= rtol*|b| + floor (for strictly positive quantities such as the
    posterior variance, where the rms term of assert_close would hide errors of small entries).
    ``floor`` is the resolution of the arithmetic itself: var = Kout - c^T K^-1 c is a difference
    of O(1) numbers, so fp32 cannot resolve it below a few ulp of Kout = 1 (pass 1e-6 there)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    err = np.abs(got - ref)
    bad = ~(err <= rtol * np.abs(ref) + floor)
    assert not bad.any(), (
        f"{what}: {bad.sum()} of {ref.size} outside relative tol {rtol} (+{floor}); "
        f"max rel err {np.nanmax(err / np.abs(ref)):.3e}"
    )


def to_dev(x, dtype=None, device="cuda"):
    import torch

    t = torch.as_tensor(np.ascontiguousarray(x), device=device)
    if dtype is not None and t.is_floating_point():
        t = t.to(dtype)
    return t
