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
