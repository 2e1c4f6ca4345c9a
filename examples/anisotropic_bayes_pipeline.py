#!/usr/bin/env python3
"""BASELINE config 4 end to end on one MI355X: anisotropic Matern-3/2 local GP, fp64, k = 50,
d = 8, leave-one-out likelihood, full Bayes-opt hyper-parameter loop, then prediction.
(BASELINE's nugget is 1e-5; the synthetic data here carry white noise of variance 1e-3, which the
model is told about.)

    python examples/anisotropic_bayes_pipeline.py [--points 1000000] [--batch 1000000] [--optimizer bayes]

Stages (each timed): synthetic data with planted per-feature length scales -> exact k-NN on the
GPU (fused MFMA scan, fp32 features) -> LOOCV batch -> optimiser over the d length scales (every
objective evaluation is ONE fused launch + a 7-scalar reduction) -> analytic sigma^2 -> posterior
mean / variance for held-out points.  ``--points 10000000`` is the BASELINE size.
"""

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from muygpys_amd.gp import MuyGPS
from muygpys_amd.gp.deformation import Anisotropy, l2
from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter, VectorParameter
from muygpys_amd.gp.kernels import Matern
from muygpys_amd.gp.noise import HomoscedasticNoise
from muygpys_amd.neighbors import NN_Wrapper
from muygpys_amd.optimize import Bayes_optimize, L_BFGS_B_optimize
from muygpys_amd.optimize.loss import lool_fn


class Clock:
    def __init__(self):
        self.rows = []

    def __call__(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.rows.append((name, time.perf_counter() - t0))
        return out


def synth(n, d, true_ls, gen, noise_std=0.03, features=2048, chunk=1 << 20):
    """An (approximate) draw from the model itself: random Fourier features of an anisotropic
    Matern-3/2 GP with the planted length scales -- frequencies ~ multivariate Student-t(3) / l_j,
    y = sqrt(2/M) sum_m cos(w_m . x + phi_m) -- plus white noise.  Exact GP sampling is impossible at
    10^6..10^7 points; this has the right covariance up to O(1/sqrt(M))."""
    X = 4.0 * torch.rand(n, d, generator=gen, device="cuda", dtype=torch.float32)
    z = torch.randn(features, d, generator=gen, device="cuda", dtype=torch.float64)
    u = (torch.randn(features, 3, generator=gen, device="cuda", dtype=torch.float64) ** 2).sum(1)  # chi^2_3
    omega = z / true_ls[None, :] * torch.sqrt(3.0 / u)[:, None]
    phi = 2.0 * np.pi * torch.rand(features, generator=gen, device="cuda", dtype=torch.float64)
    y = torch.empty(n, device="cuda", dtype=torch.float64)
    for s in range(0, n, chunk):
        y[s:s + chunk] = torch.cos(X[s:s + chunk].double() @ omega.T + phi).sum(1) * (2.0 / features) ** 0.5
    y += noise_std * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
    return X, y


def run(points=1_000_000, test_points=100_000, batch=1_000_000, d=8, k=50, optimizer="bayes", seed=0, n_iter=20,
        init_points=5, verbose=True, x0=1.0, bounds=(0.1, 10.0)):
    clock = Clock()
    gen = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(2)
    true_ls = torch.tensor(np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=d)), device="cuda")
    X32, y = clock("synthetic data", lambda: synth(points + test_points, d, true_ls, gen))
    Xtr32, Xte32, ytr, yte = X32[:points].contiguous(), X32[points:].contiguous(), y[:points].contiguous(), y[points:]
    Xtr, Xte = Xtr32.double(), Xte32.double()

    nbrs = clock("k-NN index (norms)", lambda: NN_Wrapper(Xtr32, k))
    batch = min(batch, points)
    bi = torch.randperm(points, generator=gen, device="cuda")[:batch].sort().values
    ni = clock(f"k-NN, {batch} batch rows x {points} points", lambda: nbrs.get_batch_nns(bi)[0])

    model = MuyGPS(
        kernel=Matern(
            smoothness=Parameter(1.5),
            deformation=Anisotropy(l2, length_scale=VectorParameter(*[Parameter(float(x0), tuple(bounds)) for _ in range(d)])),
        ),
        noise=HomoscedasticNoise(1e-3),  # ~ the planted noise variance (0.03^2)
        scale=AnalyticScale(),
    )
    cross, pair, y_b, y_nn = model.make_train_tensors(bi, ni, Xtr, ytr)
    evals = {"n": 0}
    # optimizers: "bayes" (the reference's defaults), "bayes-log" (the same budget, searched in the logarithms of the
    # length scales), "lbfgs" (scipy L-BFGS-B on finite differences, as in the reference), "lbfgs-analytic" (round 5:
    # one forward + one backward launch per iteration)
    opt = Bayes_optimize if optimizer.startswith("bayes") else L_BFGS_B_optimize
    if optimizer.startswith("bayes"):
        kwargs = dict(init_points=init_points, n_iter=n_iter, random_state=seed, log_bounds=optimizer == "bayes-log")
    else:
        kwargs = dict(analytic_gradient=optimizer == "lbfgs-analytic")
    launches = {"forward": 0, "backward": 0}
    import muygpys_amd.fused as F_

    real = {n_: getattr(F_, n_) for n_ in ("posterior_mean_var", "loocv_partials")}
    for n_, fn_ in real.items():
        setattr(F_, n_, (lambda fn__: lambda *a_, **k_: (launches.__setitem__("forward", launches["forward"] + 1), fn__(*a_, **k_))[1])(fn_))
    real_vg = F_.loocv_value_and_grad
    F_.loocv_value_and_grad = lambda *a_, **k_: (launches.__setitem__("backward", launches["backward"] + 1), real_vg(*a_, **k_))[1]

    def fit():
        obj = opt.make_obj_fn(model, y_b, y_nn, cross, pair, loss_fn=lool_fn, loss_kwargs={})

        def counted(*a, **kw):
            evals["n"] += 1
            return obj(*a, **kw)

        counted.loocv_context = obj.loocv_context  # (what an analytic-gradient driver differentiates)
        return opt._fn(model, counted, verbose=False, **kwargs)

    fitted = clock(f"{optimizer} optimisation over {d} length scales", fit)
    fit_launches = dict(launches)
    for n_, fn_ in real.items():
        setattr(F_, n_, fn_)
    F_.loocv_value_and_grad = real_vg
    fitted = clock("analytic sigma^2", lambda: fitted.optimize_scale(pair, y_nn))

    ti = clock(f"k-NN, {test_points} test rows", lambda: nbrs.get_nns(Xte32)[0])

    def predict():
        c, p, tn = fitted.make_predict_tensors(None, ti, Xte, Xtr, ytr)
        Kin, Kc = fitted.kernel(p), fitted.kernel(c)
        return fitted.posterior_mean(Kin, Kc, tn), fitted.posterior_variance(Kin, Kc)

    mean, var = clock("posterior mean + variance", predict)
    ls = np.asarray([float(v) for v in fitted.kernel.deformation.length_scale()])
    rmse = float(((mean - yte) ** 2).mean().sqrt())
    cover = float(((mean - yte).abs() <= 1.96 * var.sqrt()).double().mean())
    opt_s = dict(clock.rows)[f"{optimizer} optimisation over {d} length scales"]
    out = dict(points=points, batch=batch, nn_count=k, feature_count=d, optimizer=optimizer,
               objective_evaluations=evals["n"] or fit_launches["backward"],
               seconds_per_evaluation=opt_s / max(evals["n"] or fit_launches["backward"], 1),
               fused_launches_of_the_fit=fit_launches, start_length_scale=float(x0),
               median_sq_rel_error_of_length_scales=float(np.median((ls / true_ls.cpu().numpy() - 1.0) ** 2)),
               median_sq_rel_error_of_the_start=float(np.median((float(x0) / true_ls.cpu().numpy() - 1.0) ** 2)),
               length_scale=ls.round(3).tolist(), true_length_scale=true_ls.cpu().numpy().round(3).tolist(),
               sigma_sq=float(np.asarray(fitted.scale()).reshape(-1)[0]), rmse=rmse, target_std=float(yte.std()),
               coverage_95=cover, seconds={name: round(s, 4) for name, s in clock.rows})
    if verbose:
        for name, s in clock.rows:
            print(f"{name:55s} {s * 1e3:11.1f} ms")
        print({k_: v for k_, v in out.items() if k_ != "seconds"})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--test-points", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--optimizer", default="bayes", choices=["bayes", "bayes-log", "lbfgs", "lbfgs-analytic"])
    ap.add_argument("--n-iter", type=int, default=20)
    ap.add_argument("--out", default="", help="write the result (per-stage seconds included) as JSON to this file")
    a = ap.parse_args()
    res = run(points=a.points, test_points=a.test_points, batch=a.batch, optimizer=a.optimizer, n_iter=a.n_iter)
    if a.out:
        import json

        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")
