"""Run-time specialised wave kernels (csrc/mgp_jit.hip): any static shape against the oracle and against
the LDS workgroup kernel, on batches large enough for the dispatcher to choose them."""

import numpy as np
import pytest
import torch

from oracle import muygps_oracle as orc
from tests.util import RTOL, assert_close, to_dev

pytestmark = pytest.mark.gpu

B = 65536 + 7  # >= MUYGPYS_HIP_JIT_MIN_BATCH: the dispatcher takes the run-time compiled kernel; odd on purpose
N = 12000

#        k   d  R  dtype      kernel      metric  aniso  packed
SHAPES = [
    (10, 8, 1, "float32", "matern15", "l2", False, True),      # 16 slots: four neighbourhoods per wave
    (5, 4, 1, "float32", "rbf", "F2", False, True),
    (14, 12, 1, "float32", "matern25", "l2", True, False),     # 16 slots exactly, anisotropic, plain tables
    (13, 16, 2, "float32", "matern15", "l2", False, True),
    (20, 32, 1, "float32", "matern15", "l2", False, True),     # 32 slots, modulo-21 pair scheme with the Gram form
    (25, 40, 1, "float32", "maternInf", "l2", False, False),
    (25, 20, 3, "float32", "matern05", "l2", False, True),     # Matern-1/2: difference form
    (29, 64, 1, "float32", "matern15", "l2", True, True),
    (31, 8, 1, "float32", "matern15", "l2", False, True),      # 64 slots
    (40, 40, 1, "float32", "matern15", "l2", False, True),     # the reference's default nn_count
    (45, 24, 4, "float32", "rbf", "F2", False, False),
    (62, 16, 1, "float32", "matern15", "l2", False, True),     # 64 slots exactly
    # the folded elimination's corners (fp32 Gram-form shapes with k >= slots / 2; phase 4F)
    (16, 8, 1, "float32", "matern15", "l2", False, True),      # k = slots / 2: the query row is the first long row
    (16, 16, 4, "float32", "rbf", "F2", False, True),          # ... with four responses from the prepared table
    (29, 24, 2, "float32", "matern25", "l2", True, True),      # two responses, anisotropic, 32 of 32 slots
    (22, 56, 1, "float32", "matern15", "l2", False, False),    # long rows: centred four groups at a time; plain tables
    (32, 8, 1, "float32", "matern15", "l2", False, True),      # 64 slots, k = slots / 2
    (33, 12, 3, "float32", "maternInf", "l2", False, True),    # 64 slots, three responses
    (10, 8, 1, "float64", "matern15", "l2", False, True),
    (25, 16, 1, "float64", "matern25", "l2", True, True),
    (40, 8, 2, "float64", "matern15", "l2", False, False),
    (30, 32, 1, "float64", "rbf", "F2", False, True),
    # the dealt-lower-triangle elimination (fp64, 64 slots, one response; phase 4D): odd and even nn_count (row pairs
    # start on an upper-triangle element for odd columns; an odd row count pads a phantom row), a column whose pairs
    # straddle two slots at every step parity, the full 64 slots, plain and prepared tables, both deformations
    (31, 8, 1, "float64", "matern15", "l2", False, True),
    (33, 12, 1, "float64", "matern25", "l2", True, True),
    (40, 40, 1, "float64", "matern15", "l2", False, True),
    (47, 16, 1, "float64", "rbf", "F2", False, False),
    (50, 16, 1, "float64", "matern05", "l2", False, True),
    (61, 8, 1, "float64", "maternInf", "l2", True, True),
    (62, 24, 1, "float64", "matern15", "l2", False, False),
    # ... with several responses (response rows q + 1 + r of the dealt triangle; two ride in a prepared fp64 row)
    (33, 8, 2, "float64", "matern15", "l2", True, True),
    (45, 12, 3, "float64", "rbf", "F2", False, False),
    (58, 16, 4, "float64", "matern25", "l2", False, False),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"k{s[0]}_d{s[1]}_R{s[2]}_{s[3][-2:]}_{s[4]}{'_aniso' if s[6] else ''}{'_packed' if s[7] else ''}")
def test_runtime_compiled_kernels_match_oracle_and_workgroup_kernel(shape):
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    k, d, R, dtype, kernel, metric, aniso, packed = shape
    lib = _lib.load()
    if lib.mgp_jit_mode() == 0:
        pytest.skip("MUYGPYS_HIP_JIT=0")
    td = getattr(torch, dtype)
    rng = np.random.default_rng(1000 + 31 * k + d)
    X = rng.normal(size=(N, d))
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.1 * rng.normal(size=(N, R))
    y = Y[:, 0] if R == 1 else Y
    bi = rng.integers(0, N, size=B)
    ni = rng.integers(0, N - 1, size=(B, k))
    ni = ni + (ni >= bi[:, None])
    base = float(np.sqrt(d)) * (1.2 if metric == "l2" else 1.0)
    ls = list(base * np.exp(rng.uniform(-0.4, 0.4, size=d))) if aniso else base
    noise = 1e-3 if dtype == "float32" else 1e-5
    spec = KernelSpec(kernel, metric, ls, noise)
    Xd, yd, bid, nid = to_dev(X, td), to_dev(y, td), to_dev(bi), to_dev(ni)
    before = lib.mgp_jit_loaded_count()
    mean, var, yk = posterior_mean_var(spec, Xd, Xd, bid, nid, yd, want_ykinvy=True, packed=packed)
    torch.cuda.synchronize()
    assert lib.mgp_jit_loaded_count() == before + 1, "this shape must have been served by a run-time compiled kernel"
    # every neighbourhood against the LDS workgroup kernel (independent code, same precision)
    gm, gv, gy = posterior_mean_var(spec, Xd, Xd, bid, nid, yd, want_ykinvy=True, path="generic")
    torch.cuda.synchronize()
    rtol = RTOL[dtype]
    assert_close(mean.cpu().numpy(), gm.cpu().numpy(), rtol, "mean vs workgroup kernel")
    assert_close(var.cpu().numpy(), gv.cpu().numpy(), rtol, "var vs workgroup kernel")
    assert_close(yk.cpu().numpy(), gy.cpu().numpy(), rtol, "ykinvy vs workgroup kernel")
    # a sample (first, last, odd tail included) against the fp64 oracle
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(B - 40, B), rng.integers(0, B, size=120)]))
    ospec = orc.Spec(kernel, metric, np.asarray(ls) if aniso else ls, noise)
    m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi[rows], ni[rows], y)
    assert_close(mean.cpu().numpy()[rows].reshape(m_ref.shape), m_ref, rtol, "mean vs oracle")
    assert_close(var.cpu().numpy()[rows], v_ref, rtol, "var vs oracle")
    # the second call reuses the loaded kernel
    posterior_mean_var(spec, Xd, Xd, bid, nid, yd, packed=packed)
    assert lib.mgp_jit_loaded_count() == before + 1


def test_small_batches_keep_the_runtime_shape_kernels():
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    lib = _lib.load()
    if lib.mgp_jit_mode() != 1:
        pytest.skip("automatic mode only")
    gen = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn(4000, 24, device="cuda", generator=gen)
    y = torch.randn(4000, device="cuda", generator=gen)
    bi = torch.randint(0, 4000, (1000,), device="cuda", generator=gen)
    ni = torch.randint(0, 4000, (1000, 17), device="cuda", generator=gen)
    before = lib.mgp_jit_loaded_count()
    posterior_mean_var(KernelSpec("matern15", "l2", 5.0, 1e-3), X, X, bi, ni, y)
    torch.cuda.synchronize()
    assert lib.mgp_jit_loaded_count() == before


def test_medium_batches_take_cached_kernels_but_never_compile():
    """From MUYGPYS_HIP_JIT_CACHED_MIN_BATCH (4096) neighbourhoods on a call takes the specialised kernel of its
    shape when that lies in the disk cache already (the shapes compiled at build time) -- and still never waits
    for a compile below MUYGPYS_HIP_JIT_MIN_BATCH."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    lib = _lib.load()
    if lib.mgp_jit_mode() != 1:
        pytest.skip("automatic mode only")
    gen = torch.Generator(device="cuda").manual_seed(8)
    b = 8192 + 3

    def run(k, d):
        X = torch.randn(6000, d, device="cuda", generator=gen)
        y = torch.randn(6000, device="cuda", generator=gen)
        bi = torch.randint(0, 6000, (b,), device="cuda", generator=gen)
        ni = torch.randint(0, 6000, (b, k), device="cuda", generator=gen)
        spec = KernelSpec("matern15", "l2", float(np.sqrt(2 * d)), 1e-3)
        before = lib.mgp_jit_loaded_count()
        got = posterior_mean_var(spec, X, X, bi, ni, y, packed=True)
        ref = posterior_mean_var(spec, X, X, bi, ni, y, path="generic", packed=False)
        torch.cuda.synchronize()
        for g, r in zip(got, ref):
            assert_close(g.cpu().numpy(), r.cpu().numpy(), RTOL["float32"], f"k={k} d={d}")
        return lib.mgp_jit_loaded_count() - before

    assert run(25, 16) == 1, "a shape of the build-time cache must be served by its specialised kernel"
    assert run(23, 28) == 0, "a shape that would have to be compiled must not be, at this batch size"
