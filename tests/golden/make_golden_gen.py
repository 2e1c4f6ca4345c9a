#!/usr/bin/env python3
"""Generate tests/golden/gen_*.npz: the general-smoothness Matern kernel of the REAL reference
(numpy backend, fp64, scipy.special.kv): ``_matern_gen_fn`` on a grid of distances for
nu in {0.42, 1.0, 3.7, ...} (_src/gp/kernels/numpy.py:34-43; the reference pins it against
scikit-learn at nu = 0.42 in tests/kernels.py:429-526), and a MuyGPS model whose smoothness is a
free hyper-parameter (gp/kernels/matern.py:61-81 selects the general form): kernel tensors,
posterior mean / variance, sigma^2 and LOOCV objective values at smoothness probes.

Run in the build container only (it needs /root/reference, which never travels):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gen.py

Nothing from the reference is copied: this script imports MuyGPyS, feeds it seeded inputs and
stores inputs + outputs as data (import shims as in make_golden.py).
"""

import importlib.metadata as md
import json
import os
import sys
import types

_v = md.version
md.version = lambda n: "0.9.0" if n == "MuyGPyS" else _v(n)
_bo = types.ModuleType("bayes_opt")
_bo.BayesianOptimization = object
sys.modules["bayes_opt"] = _bo

import numpy as np  # noqa: E402

from MuyGPyS._src.gp.kernels.numpy import _matern_gen_fn  # noqa: E402
from MuyGPyS.gp import MuyGPS  # noqa: E402
from MuyGPyS.gp.deformation import Isotropy, l2  # noqa: E402
from MuyGPyS.gp.hyperparameter import AnalyticScale, Parameter  # noqa: E402
from MuyGPyS.gp.kernels import Matern  # noqa: E402
from MuyGPyS.gp.noise import HomoscedasticNoise  # noqa: E402
from MuyGPyS.optimize import L_BFGS_B_optimize  # noqa: E402
from MuyGPyS.optimize.loss import lool_fn, mse_fn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NUS = [0.42, 1.0, 3.7, 0.5, 1.5, 2.5, 0.05, 9.5]


def main():
    rng = np.random.default_rng(301)
    dists = np.concatenate([[0.0, 0.0], np.logspace(-9, 2.5, 120), rng.uniform(0, 6, size=(130,))]).reshape(4, 63)
    vals = np.stack([_matern_gen_fn(dists.copy(), nu) for nu in NUS])  # the reference overwrites its input
    np.savez_compressed(os.path.join(HERE, "gen_matern_function.npz"), dists=dists, smoothness=np.array(NUS), values=vals,
                        meta=np.array(json.dumps(dict(name="gen_matern_function"))))

    N, d, k, b = 300, 6, 10, 24
    X = rng.normal(size=(N, d))
    y = np.sin(X @ (rng.normal(size=d) / np.sqrt(d))) + 0.1 * rng.normal(size=N)
    batch_idx = np.sort(rng.choice(N, size=b, replace=False)).astype(np.int64)
    nn_idx = np.stack([rng.choice(np.delete(np.arange(N), i), size=k, replace=False) for i in batch_idx]).astype(np.int64)
    nu0, ls, noise = 0.42, 2.0, 1e-3
    m = MuyGPS(kernel=Matern(smoothness=Parameter(nu0, (0.1, 5.0)), deformation=Isotropy(l2, length_scale=Parameter(ls))),
               noise=HomoscedasticNoise(noise), scale=AnalyticScale())
    cross, pair, y_b, y_nn = m.make_train_tensors(batch_idx, nn_idx, X, y)
    Kin, Kc = m.kernel(pair), m.kernel(cross)
    mean = m.posterior_mean(Kin, Kc, y_nn)
    var = m.get_opt_var_fn()(Kin, Kc)
    m = m.optimize_scale(pair, y_nn)
    probes = [0.42, 1.0, 3.7, 0.8]
    obj_lool = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=lool_fn)
    obj_mse = L_BFGS_B_optimize.make_obj_fn(m, y_b, y_nn, cross, pair, loss_fn=mse_fn)
    meta = dict(name="gen_m042_iso_k10_d6", N=N, d=d, k=k, b=b, R=1, kernel="matern_gen", metric="l2", smoothness=nu0,
                length_scale=ls, noise=noise, probes=probes)
    np.savez_compressed(
        os.path.join(HERE, "gen_m042_iso_k10_d6.npz"), features=X, targets=y, batch_idx=batch_idx, nn_idx=nn_idx,
        Kin=Kin, Kcross=Kc, mean=mean, var_unscaled=var, sigma_sq=np.asarray(m.scale()).reshape(-1),
        probe_lool=np.array([float(obj_lool(smoothness=p)) for p in probes]),
        probe_mse=np.array([float(obj_mse(smoothness=p)) for p in probes]),
        meta=np.array(json.dumps(meta)),
    )
    for f in ("gen_matern_function", "gen_m042_iso_k10_d6"):
        print(f, os.path.getsize(os.path.join(HERE, f + ".npz")) // 1024, "kB")


if __name__ == "__main__":
    main()
