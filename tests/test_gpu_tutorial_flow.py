"""GPU: BASELINE configs[0] -- the univariate regression tutorial flow end to end on the hip
backend (kNN -> LOOCV batch -> L-BFGS-B on the lool objective -> analytic sigma^2 -> prediction
with uncertainty).  Statistical assertions in the spirit of the reference's tests/optimize.py
and tests/scale_opt.py: planted hyper-parameters are recovered within loose tolerances on the
MEDIAN over several sampled curves (src/MuyGPyS/_test/optimize.py:37-49 uses median relative
errors of 0.25-0.9)."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_tutorial_flow_recovers_planted_model():
    from examples.univariate_regression import run

    outs = [run(seed=s, verbose=False) for s in range(5)]
    for o in outs:
        assert o["rmse"] < 0.05 * o["target_std"], o  # interpolates the held-out points
    med = lambda key: float(np.median([o[key] for o in outs]))  # noqa: E731
    true_ls, true_s2 = outs[0]["true_length_scale"], outs[0]["true_sigma_sq"]
    assert abs(med("length_scale") - true_ls) / true_ls < 0.5
    assert abs(med("sigma_sq") - true_s2) / true_s2 < 0.9
    assert med("coverage") >= 0.75  # 95 % intervals roughly calibrated for a k = 10 local GP


def test_anisotropic_pipeline_recovers_planted_length_scales():
    """Config-4 flow in miniature (examples/anisotropic_bayes_pipeline.py): data drawn from an
    anisotropic Matern-3/2 GP (random Fourier features) with planted per-feature length scales,
    exact GPU k-NN, L-BFGS-B over the 8 length scales of the lool objective, analytic sigma^2,
    prediction.  The optimiser must land near the planted values and the predictive intervals
    must be calibrated."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "anisotropic_bayes_pipeline.py")
    spec = importlib.util.spec_from_file_location("anisotropic_bayes_pipeline", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(points=300_000, test_points=20_000, batch=60_000, optimizer="lbfgs", verbose=False)
    ls, true = np.asarray(out["length_scale"]), np.asarray(out["true_length_scale"])
    assert np.all(np.abs(ls / true - 1.0) < 0.35), (ls, true)
    assert 0.6 < out["sigma_sq"] < 1.6
    assert out["rmse"] < 0.8 * out["target_std"]
    assert 0.90 < out["coverage_95"] < 0.99
