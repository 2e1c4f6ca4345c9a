"""Pin the numpy oracle to the real reference through the committed fixtures.

The fixtures in tests/golden/ were produced by importing /root/reference (see
tests/golden/make_golden.py); here every stage of the oracle must reproduce
them to fp64 round-off.  CPU only.
"""

import json
import os

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.conftest import GOLDEN_DIR, spec_from_meta

RTOL = 1e-10
ATOL = 1e-12


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_tensors_and_kernels(golden):
    g, meta = golden, golden["meta"]
    spec = spec_from_meta(meta, g)
    X, bi, ni = g["features"], g["batch_idx"], g["nn_idx"]
    cd = orc.crosswise_tensor(X, X, bi, ni)
    pd = orc.pairwise_tensor(X, ni)
    if spec.anisotropic:
        if "crosswise" in g:
            close(cd, g["crosswise"])
        if "pairwise" in g:
            close(pd, g["pairwise"])
    else:
        if "crosswise" in g:
            close(orc.metric_reduce(cd, spec.metric), g["crosswise"])
        if "pairwise" in g:
            close(orc.metric_reduce(pd, spec.metric), g["pairwise"])
    Kc, Kin = orc.kernel_tensors(spec, cd, pd)
    close(Kc, g["Kcross"])
    if "Kin" in g:
        close(Kin, g["Kin"])
    if "Kin_perturbed" in g:
        close(orc.perturb(spec, Kin, ni), g["Kin_perturbed"])
    close(g["targets"][bi], g["batch_targets"])
    close(g["targets"][ni], g["batch_nn_targets"])


def test_posterior_and_scale(golden):
    g, meta = golden, golden["meta"]
    spec = spec_from_meta(meta, g)
    X, y, bi, ni = g["features"], g["targets"], g["batch_idx"], g["nn_idx"]
    mean, var = orc.posterior_mean_var(spec, X, X, bi, ni, y)
    close(mean, g["mean"], rtol=1e-8, atol=1e-10)
    close(var, g["var_unscaled"], rtol=1e-8, atol=1e-10)
    s = np.atleast_1d(orc.sigma_sq(spec, X, ni, y))
    close(s, g["sigma_sq"], rtol=1e-9)
    if "var_scaled" in g:
        close(s[0] * var, g["var_scaled"], rtol=1e-8, atol=1e-10)
    mc, vc = orc.posterior_mean_var_chunked(spec, X, X, bi, ni, y, chunk=5)
    close(mc, mean)
    close(vc, var)


def test_losses(golden):
    g = golden
    if "lool" not in g:
        pytest.skip("multi-response fixture: only mse is defined")
    mean, var, y, s = g["mean"], g["var_unscaled"], g["batch_targets"], float(g["sigma_sq"][0])
    close(orc.lool_fn(mean, y, var, s), g["lool"], rtol=1e-9)
    close(orc.mse_fn(mean, y), g["mse"], rtol=1e-9)
    close(orc.looph_fn(mean, y, var, s), g["looph"], rtol=1e-9)
    close(orc.pseudo_huber_fn(mean, y), g["huber"], rtol=1e-9)


def test_mse_multiresponse(golden):
    g = golden
    close(orc.mse_fn(g["mean"], g["batch_targets"]), g["mse"], rtol=1e-9)


def test_objective_probes(golden):
    g, meta = golden, golden["meta"]
    if "probe_values" not in g:
        pytest.skip("no probes in this fixture")
    X, y, bi, ni = g["features"], g["targets"], g["batch_idx"], g["nn_idx"]
    for j, probe in enumerate(meta["probes"]):
        spec = spec_from_meta(meta, g)
        if "length_scale" in probe:
            spec.length_scale = probe["length_scale"]
        elif "length_scale0" in probe:
            spec.length_scale = np.array([probe[f"length_scale{i}"] for i in range(meta["d"])])
        # App. B 3b: sigma_sq keeps the STORED noise even when `noise` is a free kwarg
        spec_mv = spec_from_meta(meta, g)
        spec_mv.length_scale = spec.length_scale
        if "noise" in probe:
            spec_mv.noise = probe["noise"]
        mean, var = orc.posterior_mean_var(spec_mv, X, X, bi, ni, y)
        s = orc.sigma_sq(spec, X, ni, y)
        yb = y[bi]
        vals = [
            -orc.lool_fn(mean, yb, var, s), -orc.mse_fn(mean, yb),
            -orc.looph_fn(mean, yb, var, s), -orc.pseudo_huber_fn(mean, yb),
        ]
        close(vals, g["probe_values"][:, j], rtol=1e-8)


def test_chunk_sizes_rule():
    with open(os.path.join(GOLDEN_DIR, "chunk_sizes.json")) as f:
        rule = json.load(f)
    for key, sizes in rule.items():
        n, p = (int(t) for t in key.split("_"))
        assert orc.chunk_sizes(n, p) == sizes
        assert sum(sizes) == n


def test_kernels_against_sklearn():
    """Known-answer pin used by the reference itself (tests/kernels.py:325-526)."""
    from sklearn.gaussian_process.kernels import RBF, Matern

    rng = np.random.default_rng(7)
    X = rng.normal(size=(40, 3))
    D = orc.pairwise_tensor(X, np.arange(40)[None, :])[0]
    ell = 1.7
    close(orc.rbf_fn(orc.isotropy(orc.F2(D), ell, "F2")), RBF(length_scale=ell)(X), rtol=1e-9)
    for name, nu in (("matern05", 0.5), ("matern15", 1.5), ("matern25", 2.5), ("maternInf", np.inf)):
        K = orc.KERNELS[name](orc.isotropy(orc.l2(D), ell, "l2"))
        close(K, Matern(length_scale=ell, nu=nu)(X), rtol=1e-9)
    ells = np.array([0.5, 1.5, 2.5])
    close(orc.matern_15_fn(orc.anisotropy(D, ells, "l2")), Matern(length_scale=ells, nu=1.5)(X), rtol=1e-9)


@pytest.mark.parametrize("name", __import__("tests.conftest", fromlist=["fast_golden_names"]).fast_golden_names())
def test_oracle_fast_posterior_mean_matches_reference(name):
    """Pins the oracle's fast-mean leg (fast_nn_update, fast_posterior_mean_precompute,
    fast_posterior_mean) to the reference's own workflow outputs (tests/golden/make_golden_fast.py:
    gp/tensors.py:52-91, gp/muygps.py:261-341, examples/fast_posterior_mean.py:72-87,374-390)."""
    from tests.conftest import load_golden, spec_from_meta

    g = load_golden(name)
    spec = spec_from_meta(g["meta"], g)
    X, y, Q = g["features"], g["targets"], g["test_features"]
    nn_fast = orc.fast_nn_update(g["train_nn"])
    assert np.array_equal(nn_fast, g["train_nn_fast"])
    pd = orc.pairwise_tensor(X, nn_fast)
    _, Kin = orc.kernel_tensors(spec, pd[:, 0], pd)
    coeffs = orc.fast_posterior_mean_precompute(orc.perturb(spec, Kin, nn_fast), y[nn_fast])
    np.testing.assert_allclose(coeffs, g["coeffs"], rtol=1e-8, atol=1e-10)
    cset = nn_fast[g["closest_neighbor"]]
    assert np.array_equal(cset, g["closest_set"])
    cd = orc.crosswise_tensor(Q, X, np.arange(len(Q)), cset)
    Kc, _ = orc.kernel_tensors(spec, cd, pd[:1])
    np.testing.assert_allclose(Kc, g["Kcross"], rtol=1e-10, atol=1e-14)
    mean = orc.fast_posterior_mean(Kc, coeffs[g["closest_neighbor"]])
    np.testing.assert_allclose(mean, g["fast_mean"], rtol=1e-8, atol=1e-10)


def test_oracle_general_matern_matches_reference():
    """K3: the oracle's general-smoothness Matern against the reference's _matern_gen_fn on a grid of
    distances (zeros included) and inside a model whose smoothness is free (make_golden_gen.py)."""
    from tests.conftest import load_golden

    g = load_golden("gen_matern_function")
    for nu, want in zip(g["smoothness"], g["values"]):
        np.testing.assert_allclose(orc.matern_gen_fn(g["dists"], float(nu)), want, rtol=1e-12, atol=0)
    g = load_golden("gen_m042_iso_k10_d6")
    meta = g["meta"]
    for nu, lool, mse in zip(meta["probes"], g["probe_lool"], g["probe_mse"]):
        spec = orc.Spec(kernel=(lambda r, nu=nu: orc.matern_gen_fn(r, nu)), metric="l2",
                        length_scale=meta["length_scale"], noise=meta["noise"])
        np.testing.assert_allclose(
            orc.loocv_objective(spec, g["features"], g["batch_idx"], g["nn_idx"], g["targets"], "lool"), lool, rtol=1e-8)
        np.testing.assert_allclose(
            orc.loocv_objective(spec, g["features"], g["batch_idx"], g["nn_idx"], g["targets"], "mse"), mse, rtol=1e-8)
        if nu == meta["smoothness"]:
            mean, var = orc.posterior_mean_var(spec, g["features"], g["features"], g["batch_idx"], g["nn_idx"], g["targets"])
            np.testing.assert_allclose(mean, g["mean"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(var, g["var_unscaled"], rtol=1e-8)


def test_fp32_reference_fixture_covers_every_forward_fixture():
    """tests/golden/fp32_reference.npz (make_golden_fp32.py: the reference's torch backend at
    MUYGPYS_FTYPE=32) calibrates the fp32 acceptance band of the HIP kernels: one (mean32, var32) pair
    per forward fixture, same shapes as the fp64 results, and never better than fp32 can be."""
    from tests.conftest import golden_names, load_golden
    from tests.util import fp32_reference

    for name in golden_names():
        g = load_golden(name)
        mean32, var32 = fp32_reference(name)
        assert mean32 is not None, name
        assert mean32.dtype == np.float32 and var32.dtype == np.float32
        assert mean32.size == g["mean"].size and var32.shape == g["var_unscaled"].shape, name
        rel = np.abs(var32.astype(np.float64) - g["var_unscaled"]) / np.abs(g["var_unscaled"])
        assert np.all(np.isfinite(rel)), name
