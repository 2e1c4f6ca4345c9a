"""The general-smoothness Matern inside the fused kernels (mgp_posterior_gen_*): K3, _matern_gen_fn
(_src/gp/kernels/numpy.py:34-43), selected whenever the smoothness is not 1/2, 3/2, 5/2 or inf
(gp/kernels/matern.py:61-81) -- e.g. while it is being optimised."""

import numpy as np
import pytest
import torch

from oracle import muygps_oracle as orc
from tests.util import assert_close, to_dev

pytestmark = pytest.mark.gpu

#         k   d  R  b      aniso packed  (b >= 65536: the run-time compiled static kernel serves the call)
SHAPES = [
    (20, 32, 1, 3000, False, True),
    (12, 16, 2, 2000, True, False),
    (10, 6, 1, 1500, False, False),     # rows not 16-byte multiples: the register-staged gather
    (30, 40, 1, 4000, False, True),     # the built-in headline instantiation
    (25, 40, 1, 70000, False, True),    # static, modulo-26 pairs
    (40, 8, 1, 66000, True, True),      # 64 slots, static
]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("nu", [0.42, 1.0, 2.2, 3.7, 8.0])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"k{s[0]}_d{s[1]}_R{s[2]}_b{s[3]}{'_aniso' if s[4] else ''}{'_packed' if s[5] else ''}")
def test_fused_general_smoothness_matches_oracle(shape, nu, dtype):
    """fp32: the hardware-exp trapezoid in every wave kernel; fp64 (round 4): the software-exp form in the GEN64
    instantiations (run-time-shape 32-slot kernels built in, static shapes compiled at run time), held to the
    north_star's 1e-5."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, d, R, b, aniso, packed = shape
    N = 9000
    rng = np.random.default_rng(77 + k + d)
    X = rng.normal(size=(N, d))
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.1 * rng.normal(size=(N, R))
    y = Y[:, 0] if R == 1 else Y
    bi = rng.integers(0, N, size=b)
    ni = rng.integers(0, N - 1, size=(b, k))
    ni = ni + (ni >= bi[:, None])
    ni[::97, 1] = ni[::97, 0]  # duplicated neighbours: k(0) = 1 exactly, and the node count of their wave grows
    base = float(np.sqrt(d)) * 1.3
    ls = list(base * np.exp(rng.uniform(-0.3, 0.3, size=d))) if aniso else base
    spec = KernelSpec("matern_gen", "l2", ls, 2e-3, smoothness=nu)
    td = getattr(torch, dtype)
    rtol = 1e-3 if dtype == "float32" else 1e-5
    mean, var, yk = posterior_mean_var(spec, to_dev(X, td), to_dev(X, td), to_dev(bi), to_dev(ni), to_dev(y, td),
                                       want_ykinvy=True, packed=packed)  # raises FusedUnsupported if not fused
    torch.cuda.synchronize()
    if dtype == "float64":
        assert "gen64" in _lib.last_kernel(), _lib.last_kernel()
    rows = np.unique(np.concatenate([np.arange(0, 200, 1), np.arange(b - 64, b), np.arange(0, b, 97)[:60]]))
    ospec = orc.Spec(lambda r: orc.matern_gen_fn(r, nu), "l2", np.asarray(ls) if aniso else ls, 2e-3)
    m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi[rows], ni[rows], y)
    assert_close(mean.cpu().numpy()[rows].reshape(m_ref.shape), m_ref, rtol, "mean")
    assert_close(var.cpu().numpy()[rows], v_ref, rtol, "var")
    assert torch.isfinite(mean).all() and torch.isfinite(var).all() and torch.isfinite(yk).all()


def test_fp64_free_smoothness_is_fused_and_matches_the_reference_fixture():
    """fp64 tables (the reference's default precision) with a free smoothness are served by the fused kernels since
    round 4 (GEN64 instantiations) -- against the reference-generated model with nu = 0.42
    (tests/golden/make_golden_gen.py); a 64-slot shape in a batch too small to compile for still raises
    FusedUnsupported (the caller materialises, as before)."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import FusedUnsupported, KernelSpec, posterior_mean_var
    from tests.conftest import load_golden

    g = load_golden("gen_m042_iso_k10_d6")
    meta = g["meta"]
    X, y = to_dev(g["features"], torch.float64), to_dev(g["targets"], torch.float64)
    spec = KernelSpec("matern_gen", meta["metric"], meta["length_scale"], meta["noise"], smoothness=meta["smoothness"])
    mean, var, yk = posterior_mean_var(spec, X, X, to_dev(g["batch_idx"]), to_dev(g["nn_idx"]), y, want_ykinvy=True)
    torch.cuda.synchronize()
    assert "gen64" in _lib.last_kernel(), _lib.last_kernel()
    assert_close(mean.cpu().numpy(), g["mean"], 1e-5, "mean")
    assert_close(var.cpu().numpy(), g["var_unscaled"], 1e-5, "var")
    b, k = g["nn_idx"].shape
    assert_close(yk.double().sum().cpu().numpy().reshape(-1) / (b * k), np.asarray(g["sigma_sq"]).reshape(-1), 1e-5, "sigma_sq")
    gen = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(500, 8, device="cuda", dtype=torch.float64, generator=gen)
    y = torch.randn(500, device="cuda", dtype=torch.float64, generator=gen)
    bi = torch.arange(50, device="cuda")
    ni = torch.randint(50, 500, (50, 40), device="cuda", generator=gen)
    with pytest.raises(FusedUnsupported):
        posterior_mean_var(KernelSpec("matern_gen", "l2", 2.0, 1e-3, smoothness=0.7), X, X, bi, ni, y)


def test_free_smoothness_objective_is_one_fused_launch_and_close_to_fixed_smoothness_speed(capsys):
    """The point of the exercise: an objective evaluation with a FREE smoothness (the general Bessel form)
    at the BASELINE config-2 shape, against the same evaluation at the fixed nu = 3/2 closed form.  Before
    round 3 the free form ran on materialised (b, k, k) distances through a 64 GB/s fp64 Bessel kernel."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    gen = torch.Generator(device="cuda").manual_seed(5)
    N, d, k, b = 400_000, 40, 30, 400_000
    X = torch.randn(N, d, device="cuda", generator=gen)
    y = torch.randn(N, device="cuda", generator=gen)
    bi = torch.arange(b, device="cuda")
    ni = torch.randint(0, N - 1, (b, k), device="cuda", generator=gen)
    ni = ni + (ni >= bi[:, None])
    times = {}
    for name, spec in (("fixed nu=3/2", KernelSpec("matern15", "l2", 6.0, 1e-3)),
                       ("free  nu=1.5 (general form)", KernelSpec("matern_gen", "l2", 6.0, 1e-3, smoothness=1.5)),
                       ("free  nu=0.8 (general form)", KernelSpec("matern_gen", "l2", 6.0, 1e-3, smoothness=0.8))):
        out = None
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = posterior_mean_var(spec, X, X, bi, ni, y, packed=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        times[name] = (float(np.median(ts[2:])), out)
    fixed, free = times["fixed nu=3/2"], times["free  nu=1.5 (general form)"]
    # nu = 3/2 through the general form is the same function: the two launches must agree
    assert_close(free[1][0].cpu().numpy(), fixed[1][0].cpu().numpy(), 1e-3, "mean, general form at nu = 3/2")
    assert_close(free[1][1].cpu().numpy(), fixed[1][1].cpu().numpy(), 1e-3, "var, general form at nu = 3/2")
    with capsys.disabled():
        print("\n[general-nu] " + "; ".join(f"{n}: {t:.3f} ms" for n, (t, _) in times.items()) + f" per {b} neighbourhoods")
    assert free[0] <= 3.0 * fixed[0], times
