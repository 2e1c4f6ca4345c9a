#!/usr/bin/env python3
"""Generate tests/golden/fast_*.npz by running the REAL reference's fast-posterior-mean workflow
(numpy backend, fp64): ``fast_nn_update`` -> ``deformation.pairwise_tensor`` -> ``kernel`` -> ``MuyGPS.fast_coefficients``
-> ``deformation.crosswise_tensor`` -> ``kernel`` -> ``MuyGPS.fast_posterior_mean`` (gp/tensors.py:52-91,
gp/muygps.py:261-341, examples/fast_posterior_mean.py:72-87,374-390, examples/from_indices.py:101-117).

Run in the build container only (it needs /root/reference, which never travels):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fast.py

Nothing from the reference is copied: this script imports MuyGPyS, feeds it seeded inputs and
stores inputs + outputs as data.  Import shims as in make_golden.py.
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

from MuyGPyS.gp import MuyGPS  # noqa: E402
from MuyGPyS.gp.deformation import Anisotropy, F2, Isotropy, l2  # noqa: E402
from MuyGPyS.gp.hyperparameter import Parameter, VectorParameter  # noqa: E402
from MuyGPyS.gp.kernels import RBF, Matern  # noqa: E402
from MuyGPyS.gp.noise import HomoscedasticNoise  # noqa: E402
from MuyGPyS.gp.tensors import fast_nn_update  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NU = {"matern05": 0.5, "matern15": 1.5, "matern25": 2.5, "maternInf": np.inf}

CASES = {
    "fast_m15_iso_k10_d8": dict(seed=201, N=400, M=60, d=8, k=10, R=1, kernel="matern15", metric="l2",
                                length_scale=2.5, noise=1e-2),
    "fast_rbf_iso_k30_d40": dict(seed=202, N=500, M=40, d=40, k=30, R=1, kernel="rbf", metric="F2",
                                 length_scale=4.0, noise=1e-2),
    "fast_m25_aniso_k12_d5_R3": dict(seed=203, N=350, M=50, d=5, k=12, R=3, kernel="matern25", metric="l2",
                                     length_scale=[1.0, 2.0, 0.7, 1.5, 3.0], noise=1e-2),
}


def knn(X, Q, k, drop_self):
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(n_neighbors=k + (1 if drop_self else 0), algorithm="brute").fit(X)
    idx = nn.kneighbors(Q, return_distance=False)
    return idx[:, 1:] if drop_self else idx


def make_case(name, c):
    rng = np.random.default_rng(c["seed"])
    N, M, d, k, R = c["N"], c["M"], c["d"], c["k"], c["R"]
    X = rng.normal(size=(N, d))
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.05 * rng.normal(size=(N, R))
    targets = Y[:, 0] if R == 1 else Y
    Q = rng.normal(size=(M, d))
    metric = l2 if c["metric"] == "l2" else F2
    ls = c["length_scale"]
    if np.ndim(ls) == 1:
        deformation = Anisotropy(metric, length_scale=VectorParameter(*[Parameter(float(v)) for v in ls]))
    else:
        deformation = Isotropy(metric, length_scale=Parameter(float(ls)))
    kernel = RBF(deformation=deformation) if c["kernel"] == "rbf" else Matern(
        smoothness=Parameter(NU[c["kernel"]]), deformation=deformation)
    m = MuyGPS(kernel=kernel, noise=HomoscedasticNoise(c["noise"]))

    train_nn = knn(X, X, k, drop_self=True).astype(np.int64)          # NN_Wrapper.get_batch_nns
    # examples/fast_posterior_mean.py:72-87 (make_fast_regressor)
    train_nn_fast = fast_nn_update(train_nn)
    nn_targets_fast = targets[train_nn_fast]
    Kin = m.kernel(m.kernel.deformation.pairwise_tensor(X, train_nn_fast))
    coeffs = m.fast_coefficients(Kin, nn_targets_fast)
    # :374-390 + examples/from_indices.py:101-117 (prediction)
    closest_neighbor = knn(X, Q, 1, drop_self=False)[:, 0].astype(np.int64)
    closest_set = train_nn_fast[closest_neighbor]
    cross = m.kernel.deformation.crosswise_tensor(Q, X, np.arange(M), closest_set)
    Kcross = m.kernel(cross)
    mean = m.fast_posterior_mean(Kcross, coeffs[closest_neighbor])
    meta = dict(c)
    meta["name"] = name
    out = dict(features=X, targets=targets, test_features=Q, train_nn=train_nn, train_nn_fast=train_nn_fast,
               coeffs=coeffs, closest_neighbor=closest_neighbor, closest_set=closest_set, Kcross=Kcross,
               fast_mean=mean, meta=np.array(json.dumps(meta)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} kB, coeffs {coeffs.shape}, mean {np.shape(mean)}")


if __name__ == "__main__":
    for name, c in CASES.items():
        make_case(name, c)
