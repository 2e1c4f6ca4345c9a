"""Pin the oracle's reverse sweep (posterior_vjp) to the reference's torch-backend autograd
through tests/golden/grad_*.npz (generator: tests/golden/make_golden_grad.py), and to finite
differences of the oracle's own forward.  CPU only."""

import numpy as np

from oracle import muygps_oracle as orc
from tests.conftest import grad_spec


def _rel(a, b):
    scale = np.abs(b).max()
    return np.abs(np.asarray(a) - np.asarray(b)).max() / (scale if scale > 0 else 1.0)


def test_forward_matches_reference_torch_backend(grad_golden):
    g, meta = grad_golden, grad_golden["meta"]
    spec, xq = grad_spec(meta, g)
    mean, var = orc.posterior_mean_var(spec, xq, g["features"], g["batch_indices"], g["nn_indices"], g["targets"])
    assert _rel(mean.reshape(g["mean"].shape), g["mean"]) < 1e-7
    assert _rel(var, g["var"]) < 1e-7


def test_vjp_matches_reference_autograd(grad_golden):
    g, meta = grad_golden, grad_golden["meta"]
    spec, xq = grad_spec(meta, g)
    out = orc.posterior_vjp(spec, xq, g["features"], g["batch_indices"], g["nn_indices"], g["targets"],
                            g["grad_mean"], g["grad_var"])
    tol = 1e-6  # the fixtures' own conditioning (nugget 1e-4..1e-2) bounds agreement of two fp64 solvers
    if meta["separate_test"]:
        assert _rel(out["test_features"], g["g_test_features"]) < tol
        assert _rel(out["train_features"], g["g_features"]) < tol
    else:
        assert _rel(out["train_features"] + out["test_features"], g["g_features"]) < tol
    assert _rel(out["targets"].reshape(g["g_targets"].shape), g["g_targets"]) < tol
    assert _rel(out["length_scale"], g["g_length_scale"]) < tol
    if meta["hetero"]:
        assert _rel(out["noise"], g["g_noise_table"]) < tol
    else:
        assert _rel(out["noise"], g["g_noise"]) < tol


def test_vjp_matches_finite_differences():
    rng = np.random.default_rng(11)
    n, d, k, b, R = 30, 3, 5, 6, 2
    X = rng.normal(size=(n, d))
    Y = rng.normal(size=(n, R))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    gm, gv = rng.normal(size=(b, R)), rng.normal(size=b)
    ls = np.array([0.9, 1.4, 0.7])

    def loss(Xv, lsv, eps):
        spec = orc.Spec("matern25", "l2", lsv, eps)
        m, v = orc.posterior_mean_var(spec, Xv, Xv, bi, ni, Y)
        return float((m * gm).sum() + (v * gv).sum())

    spec = orc.Spec("matern25", "l2", ls, 1e-2)
    out = orc.posterior_vjp(spec, X, X, bi, ni, Y, gm, gv)
    gx = out["train_features"] + out["test_features"]
    h = 1e-6
    for (i, c) in [(int(ni[0, 0]), 0), (int(bi[2]), 1), (int(ni[3, 2]), 2)]:
        Xp, Xm = X.copy(), X.copy()
        Xp[i, c] += h
        Xm[i, c] -= h
        fd = (loss(Xp, ls, 1e-2) - loss(Xm, ls, 1e-2)) / (2 * h)
        assert abs(fd - gx[i, c]) < 1e-5 * max(1.0, abs(fd))
    for c in range(d):
        lp, lm = ls.copy(), ls.copy()
        lp[c] += h
        lm[c] -= h
        fd = (loss(X, lp, 1e-2) - loss(X, lm, 1e-2)) / (2 * h)
        assert abs(fd - out["length_scale"][c]) < 1e-5 * max(1.0, abs(fd))
    fd = (loss(X, ls, 1e-2 + h) - loss(X, ls, 1e-2 - h)) / (2 * h)
    assert abs(fd - out["noise"]) < 1e-5 * max(1.0, abs(fd))
