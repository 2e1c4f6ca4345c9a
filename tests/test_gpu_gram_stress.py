"""Adversarial inputs for the fp32 Gram form of the squared distances (|a'|^2 + |b'|^2 - 2 a'.b' on rows centred on
the query; reference: difference form, src/MuyGPyS/_src/gp/tensors/numpy.py:47-94).

The Gram form cancels log2((|a'|^2 + |b'|^2) / d^2) bits.  Mild for k-NN or random neighbourhoods -- the neighbours lie
about as far from each other as from the query -- but a query far outside a tight cluster of neighbours leaves the
covariances with ~1e-5 absolute error and, at nugget 1e-3, the posterior mean with 1e-3 .. 6e-3 (the same arithmetic
restated in numpy float32; the difference form on the same data: 2e-5 .. 4e-5).  The kernels therefore carry a
cancellation guard (mgp_wave_common.h: gram_finish2; wave kernels phase 2G; rhs-column kernel): a task in which some
pair's squared distance comes out below 1/32 of the norms it was subtracted from recomputes its distances in the
difference form.  These tests fail without the guard (cases `cluster*`) and pass with it; the other cases pin that the
Gram form itself is sound where it is used."""

import zlib

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _case(name, rng, k, d, R, b):
    """-> X (n, d), Q (m, d), bi (b,), ni (b, k), length scale; all float32-representable."""
    ell = float(np.sqrt(2 * d))
    if name == "offset":  # (i) a large common offset and a separate query table
        n, m = 6000, 700
        X = 1e3 + rng.normal(size=(n, d))
        Q = 1e3 + rng.normal(size=(m, d))
        bi = rng.integers(0, m, size=b)
        ni = np.stack([rng.choice(n, size=k, replace=False) for _ in range(b)])
    elif name in ("cluster1e-1", "cluster1e-2"):  # (ii) tight cluster of neighbours 3 length scales from the query
        sig = float(name[len("cluster"):])
        Q = rng.normal(size=(b, d))
        C = rng.normal(size=(b, d))
        C *= (3 * ell / np.linalg.norm(C, axis=1))[:, None]
        X = (Q[:, None, :] + C[:, None, :] + sig * ell * rng.normal(size=(b, k, d)) / np.sqrt(d)).reshape(b * k, d)
        bi = np.arange(b)
        ni = np.arange(b * k).reshape(b, k)
    elif name == "mixed":  # every 7th neighbourhood is a far cluster, the rest random rows: the guard is per task
        n = 6000
        X0 = rng.normal(size=(n, d))
        Q = rng.normal(size=(b, d))
        C = rng.normal(size=(b, d))
        C *= (3 * ell / np.linalg.norm(C, axis=1))[:, None]
        far = (Q[:, None, :] + C[:, None, :] + 0.02 * ell * rng.normal(size=(b, k, d)) / np.sqrt(d)).reshape(b * k, d)
        X = np.concatenate([X0, far])
        bi = np.arange(b)
        ni = np.stack([rng.choice(n, size=k, replace=False) for _ in range(b)])
        sel = np.arange(b) % 7 == 3
        ni[sel] = n + np.arange(b * k).reshape(b, k)[sel]
    elif name == "near_duplicates":  # (iii) pairs of rows 1e-3 length scales apart, nugget 1e-3
        n = 6000
        X = rng.normal(size=(n, d))
        X[1::2] = X[0::2] + 1e-3 * ell * rng.normal(size=(n // 2, d)) / np.sqrt(d)
        Q = X
        bi = rng.integers(0, n, size=b)
        ni = np.stack([np.concatenate([p, p + 1])[:k] for p in (2 * rng.choice(n // 2, (k + 1) // 2, replace=False) for _ in range(b))])
        ni = np.where(ni == bi[:, None], (ni + 2) % n, ni)
    else:
        raise AssertionError(name)
    X = X.astype(np.float32).astype(np.float64)
    Q = Q.astype(np.float32).astype(np.float64)
    return X, Q, bi, ni, ell


SHAPES = [
    # k, d, R, path -- the built-in headline instantiation (folded elimination), run-time-shape 32- and 64-slot wave
    # kernels, the built-in 64-slot shape, the rhs-column kernel with and without the back-substitution variant
    (30, 40, 1, "auto"), (29, 40, 1, "auto"), (20, 16, 2, "auto"), (45, 24, 1, "auto"), (50, 8, 1, "auto"),
    (64, 40, 16, "auto"), (40, 32, 3, "rhs"),
]


@pytest.mark.parametrize("case", ["offset", "cluster1e-1", "cluster1e-2", "mixed", "near_duplicates"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"k{s[0]}-d{s[1]}-R{s[2]}-{s[3]}")
@pytest.mark.parametrize("packed", [False, True])
def test_gram_form_under_cancellation(case, shape, packed):
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, PackedTable, posterior_mean_var

    k, d, R, path = shape
    if packed and (path != "auto" or not PackedTable.supported(d, R, k, torch.float32)):
        pytest.skip("shape outside the prepared-table kernels")
    rng = np.random.default_rng(zlib.crc32(f"{case}/{k}/{d}".encode()))  # (str hashes are salted per process)
    b = 350
    X, Q, bi, ni, ell = _case(case, rng, k, d, R, b)
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d) - X.mean()) + 0.05 * rng.normal(size=(X.shape[0], R))
    spec_o = orc.Spec("matern15", "l2", ell, 1e-3)
    m_ref, v_ref = orc.posterior_mean_var(spec_o, Q, X, bi, ni, Y)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    mean, var = posterior_mean_var(KernelSpec("matern15", "l2", ell, 1e-3), to_dev(Q, torch.float32), to_dev(X, torch.float32),
                                   to_dev(bi), to_dev(ni), to_dev(Y, torch.float32), info=info, path=path, packed=packed)
    torch.cuda.synchronize()
    served = _lib.last_kernel()
    assert int(info.item()) == 0, served
    assert_close(mean.cpu().numpy().reshape(b, R), m_ref.reshape(b, R), RTOL["float32"], f"mean [{case}; {served}]")
    assert_close(var.cpu().numpy(), v_ref, RTOL["float32"], f"var [{case}; {served}]")


@pytest.mark.parametrize("case", ["cluster1e-2", "mixed"])
def test_gram_guard_in_a_long_launch_and_in_run_time_compiled_kernels(case):
    """The same through the persistent loop (every workgroup loops over many tasks; the folded elimination pairs a
    guarded task with an unguarded one) and through a run-time compiled static shape."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    for k, d in ((30, 40), (20, 8)):  # (a shape test_gpu_jit.py does not count loads of)
        rng = np.random.default_rng(7 + k)
        b = 70_000 if case == "mixed" else 12_000
        X, Q, bi, ni, ell = _case(case, rng, k, d, 1, b)
        y = np.sin(X[:, :3].sum(1) - 3 * X.mean())
        mean, var = posterior_mean_var(KernelSpec("matern15", "l2", ell, 1e-3), to_dev(Q, torch.float32),
                                       to_dev(X, torch.float32), to_dev(bi), to_dev(ni), to_dev(y, torch.float32))
        torch.cuda.synchronize()
        served = _lib.last_kernel()
        pick = np.unique(np.concatenate([np.arange(3, b, 7)[:400], rng.choice(b, size=600, replace=False), [0, 1, b - 2, b - 1]]))
        m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern15", "l2", ell, 1e-3), Q, X, bi[pick], ni[pick], y)
        assert_close(mean.cpu().numpy()[pick], m_ref, RTOL["float32"], f"mean [{served}]")
        assert_close(var.cpu().numpy()[pick], v_ref, RTOL["float32"], f"var [{served}]")
        assert bool(torch.isfinite(mean).all()) and bool(torch.isfinite(var).all())


def test_gram_form_on_real_knn_neighbourhoods_of_clustered_data():
    """(iv) exact k-NN neighbourhoods on clustered data (a mixture of tight and loose clusters plus outlying queries):
    what the pipeline actually feeds the kernels."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var
    from muygpys_amd.neighbors import NN_Wrapper

    rng = np.random.default_rng(11)
    d, k, n, m = 16, 30, 30_000, 3_000
    centres = rng.normal(size=(40, d)) * 4.0
    scale = 10.0 ** rng.uniform(-2.0, 0.0, size=40)
    which = rng.integers(0, 40, size=n)
    X = (centres[which] + scale[which, None] * rng.normal(size=(n, d))).astype(np.float32).astype(np.float64)
    Q = np.concatenate([centres[rng.integers(0, 40, size=m // 2)] + 0.5 * rng.normal(size=(m // 2, d)),
                        rng.normal(size=(m - m // 2, d)) * 8.0]).astype(np.float32).astype(np.float64)  # half of them outliers
    y = np.sin(X[:, :4].sum(1))
    Xd, Qd = to_dev(X, torch.float32), to_dev(Q, torch.float32)
    ni = NN_Wrapper(Xd, k).get_nns(Qd)[0]
    ell = 3.0
    mean, var = posterior_mean_var(KernelSpec("matern25", "l2", ell, 1e-3), Qd, Xd, None, ni, to_dev(y, torch.float32))
    torch.cuda.synchronize()
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern25", "l2", ell, 1e-3), Q, X, np.arange(m), ni.cpu().numpy(), y)
    assert_close(mean.cpu().numpy(), m_ref, RTOL["float32"], "mean")
    assert_close(var.cpu().numpy(), v_ref, RTOL["float32"], "var")
