#!/usr/bin/env python3
"""The reference's univariate regression tutorial flow (docs/examples/
univariate_regression_tutorial.ipynb: sample a curve from a GP, keep a sparse training set,
optimise the kernel on a LOOCV batch, train sigma^2, predict with uncertainty) on the hip backend.

    python examples/univariate_regression.py          # needs a ROCm device

Everything between the features and the posterior runs in the HIP kernels; the only host code is
this script and the (scalar) optimiser loop.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from muygpys_amd.gp import MuyGPS
from muygpys_amd.gp.deformation import F2, Isotropy
from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
from muygpys_amd.gp.kernels import RBF
from muygpys_amd.gp.noise import HomoscedasticNoise
from muygpys_amd.neighbors import NN_Wrapper
from muygpys_amd.optimize import L_BFGS_B_optimize
from muygpys_amd.optimize.loss import lool_fn


def sample_gp_curve(n, length_scale, sigma_sq, noise, rng):
    """Exact GP draw on a 1-D grid (the role of the reference's UnivariateSampler)."""
    x = np.linspace(0.0, 1.0, n)
    K = sigma_sq * np.exp(-((x[:, None] - x[None, :]) ** 2) / (2.0 * length_scale**2))
    L = np.linalg.cholesky(K + 1e-10 * np.eye(n))
    f = L @ rng.standard_normal(n)
    return x, f + np.sqrt(noise) * rng.standard_normal(n)


def run(seed=0, n=1100, train_step=10, nn_count=10, true_ls=0.05, true_sigma_sq=1.3, noise=1e-5,
        batch_count=200, verbose=True):
    rng = np.random.default_rng(seed)
    x, y = sample_gp_curve(n, true_ls, true_sigma_sq, noise, rng)
    test_mask = np.ones(n, dtype=bool)
    test_mask[::train_step] = False
    train_mask = ~test_mask                      # sparse training set, dense held-out test set
    dev = torch.device("cuda")
    Xtr = torch.tensor(x[train_mask], device=dev)[:, None]
    ytr = torch.tensor(y[train_mask], device=dev)
    Xte = torch.tensor(x[test_mask], device=dev)[:, None]
    yte = y[test_mask]

    muygps = MuyGPS(
        kernel=RBF(deformation=Isotropy(F2, length_scale=Parameter(0.5 * true_ls, (0.2 * true_ls, 5.0 * true_ls)))),
        noise=HomoscedasticNoise(noise),
        scale=AnalyticScale(),
    )
    nbrs = NN_Wrapper(Xtr, nn_count)
    batch_count = min(batch_count, Xtr.shape[0])
    batch_idx = torch.tensor(np.sort(rng.choice(Xtr.shape[0], batch_count, replace=False)), device=dev)
    batch_nn, _ = nbrs.get_batch_nns(batch_idx)
    cross, pair, y_b, y_nn = muygps.make_train_tensors(batch_idx, batch_nn, Xtr, ytr)
    muygps = L_BFGS_B_optimize(muygps, y_b, y_nn, cross, pair, loss_fn=lool_fn)
    muygps = muygps.optimize_scale(pair, y_nn)

    test_nn, _ = nbrs.get_nns(Xte)
    cross, pair, y_nn = muygps.make_predict_tensors(None, test_nn, Xte, Xtr, ytr)
    Kin, Kcross = muygps.kernel(pair), muygps.kernel(cross)
    mean = muygps.posterior_mean(Kin, Kcross, y_nn).cpu().numpy()
    var = muygps.posterior_variance(Kin, Kcross).cpu().numpy()
    rmse = float(np.sqrt(np.mean((mean - yte) ** 2)))
    coverage = float(np.mean(np.abs(mean - yte) <= 1.96 * np.sqrt(var)))
    out = dict(length_scale=muygps.kernel.deformation.length_scale(), sigma_sq=muygps.scale(), rmse=rmse,
               coverage=coverage, true_length_scale=true_ls, true_sigma_sq=true_sigma_sq,
               target_std=float(np.std(yte)))
    if verbose:
        print(out)
    return out


if __name__ == "__main__":
    run()
