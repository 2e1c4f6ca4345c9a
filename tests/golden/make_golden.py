#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (numpy backend, fp64).

Run in the build container only (it needs /root/reference, which never travels):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_golden.py

Nothing from the reference is copied: this script *imports* MuyGPyS, feeds it
seeded inputs, and stores inputs + outputs (every stage of the path) as data.
Import shims: package metadata for the un-installed dist and
a stub ``bayes_opt`` module (third-party, not installed; only imported, never
called here).

Each fixture holds: inputs (features, targets, indices, hyper-parameters as a
JSON string in ``meta``) and the reference's outputs ``crosswise``, ``pairwise``
(omitted when large), ``Kin``, ``Kcross``, ``Kin_perturbed``, ``mean``,
``var_scaled``, ``var_unscaled``, ``sigma_sq``, ``lool``, ``mse``, ``looph``,
``huber`` and objective values at several hyper-parameter probes.
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
from MuyGPyS.gp.hyperparameter import AnalyticScale, Parameter, VectorParameter  # noqa: E402
from MuyGPyS.gp.kernels import RBF, Matern  # noqa: E402
from MuyGPyS.gp.noise import HeteroscedasticNoise, HomoscedasticNoise  # noqa: E402
from MuyGPyS.gp.tensors import make_heteroscedastic_tensor  # noqa: E402
from MuyGPyS.optimize import L_BFGS_B_optimize  # noqa: E402
from MuyGPyS.optimize.loss import looph_fn, lool_fn, mse_fn, pseudo_huber_fn  # noqa: E402
from MuyGPyS._src.mpi_utils import _get_chunk_sizes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BIG = 150_000  # bytes: intermediates larger than this are not stored

NU = {"matern05": 0.5, "matern15": 1.5, "matern25": 2.5, "maternInf": np.inf}


def knn_indices(X, Q, k, drop_self):
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(n_neighbors=k + (1 if drop_self else 0), algorithm="brute").fit(X)
    idx = nn.kneighbors(Q, return_distance=False)
    return idx[:, 1:] if drop_self else idx


def build_model(c, length_scale, noise_obj, bounds=None):
    metric = l2 if c["metric"] == "l2" else F2
    bnd = bounds or {}

    def P(name, val):
        return Parameter(val, bnd[name]) if name in bnd else Parameter(val)

    if np.ndim(length_scale) == 1:
        deformation = Anisotropy(
            metric,
            length_scale=VectorParameter(*[P(f"length_scale{i}", float(v)) for i, v in enumerate(length_scale)]),
        )
    else:
        deformation = Isotropy(metric, length_scale=P("length_scale", float(length_scale)))
    if c["kernel"] == "rbf":
        kernel = RBF(deformation=deformation)
    else:
        kernel = Matern(smoothness=Parameter(NU[c["kernel"]]), deformation=deformation)
    return MuyGPS(kernel=kernel, noise=noise_obj, scale=AnalyticScale())


def make_case(name, c):
    rng = np.random.default_rng(c["seed"])
    N, d, k, b, R = c["N"], c["d"], c["k"], c["b"], c["R"]
    X = rng.normal(size=(N, d)) * c.get("spread", 1.0)
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.1 * rng.normal(size=(N, R))
    targets = Y[:, 0] if R == 1 else Y  # 1-D targets for R=1
    batch_idx = np.sort(rng.choice(N, size=b, replace=False)).astype(np.int64)
    if c.get("knn", False):
        nn_idx = knn_indices(X, X[batch_idx], k, drop_self=True).astype(np.int64)
    else:
        # random distinct neighbours that exclude the batch point itself
        nn_idx = np.empty((b, k), dtype=np.int64)
        for i, bi in enumerate(batch_idx):
            pool = np.delete(np.arange(N), bi)
            nn_idx[i] = rng.choice(pool, size=k, replace=False)

    ls = c["length_scale"]
    hetero = c.get("hetero", False)
    out = dict(features=X, targets=targets, batch_idx=batch_idx, nn_idx=nn_idx)
    if hetero:
        noise_table = 10.0 ** rng.uniform(-4, -1, size=N)
        out["noise_table"] = noise_table
        noise_obj = HeteroscedasticNoise(make_heteroscedastic_tensor(noise_table, nn_idx))
    else:
        noise_obj = HomoscedasticNoise(c["noise"])
    m = build_model(c, ls, noise_obj)

    cross, pair, y_b, y_nn = m.make_train_tensors(batch_idx, nn_idx, X, targets)
    Kin, Kc = m.kernel(pair), m.kernel(cross)
    Kin_p = m.noise.perturb(Kin)
    mean = m.posterior_mean(Kin, Kc, y_nn)
    var_unscaled = m.get_opt_var_fn()(Kin, Kc)
    if R == 1:
        m = m.optimize_scale(pair, y_nn)
        sigma_sq = np.asarray(m.scale(), dtype=np.float64).reshape(-1)
        var_scaled = m.posterior_variance(Kin, Kc)
        s = float(sigma_sq[0])
        out.update(
            lool=lool_fn(mean, y_b, var_unscaled, s),
            mse=mse_fn(mean, y_b),
            looph=looph_fn(mean, y_b, var_unscaled, s),
            huber=pseudo_huber_fn(mean, y_b),
            var_scaled=var_scaled,
        )
    else:
        # the reference's _analytic_scale_optim rejects R>1 (scale/numpy.py:30); the
        # per-response value is obtained the way gp/multivariate_muygps.py:375-382
        # does it: one single-response model per column.
        sig = []
        for r in range(R):
            mr = build_model(c, ls, noise_obj)
            mr = mr.optimize_scale(pair, y_nn[:, :, r])
            sig.append(float(np.asarray(mr.scale()).reshape(-1)[0]))
        sigma_sq = np.array(sig)
        out.update(mse=mse_fn(mean, y_b))
    out.update(
        batch_targets=y_b, batch_nn_targets=y_nn, Kcross=Kc, mean=mean,
        var_unscaled=var_unscaled, sigma_sq=sigma_sq,
    )
    for key, val in (("crosswise", cross), ("pairwise", pair), ("Kin", Kin), ("Kin_perturbed", Kin_p)):
        if val.nbytes <= BIG:
            out[key] = val

    # objective probes (optimize/objective.py:101-103) at fixed hyper-parameter dicts
    probes = c.get("probes")
    if probes and R == 1:
        bounds = {kname: (1e-6, 1e6) for p in probes for kname in p if kname != "noise"}
        noise_free = any("noise" in p for p in probes)
        nobj = HomoscedasticNoise(c["noise"], (1e-8, 1e2)) if (noise_free and not hetero) else noise_obj
        mo = build_model(c, ls, nobj, bounds=bounds)
        vals = {}
        for lname, lfn in (("lool", lool_fn), ("mse", mse_fn), ("looph", looph_fn), ("huber", pseudo_huber_fn)):
            obj = L_BFGS_B_optimize.make_obj_fn(mo, y_b, y_nn, cross, pair, loss_fn=lfn)
            vals[lname] = [float(obj(**p)) for p in probes]
        out["probe_values"] = np.array([vals[n] for n in ("lool", "mse", "looph", "huber")])
    meta = dict(c)
    meta["name"] = name
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} kB")


CASES = {
    "m15_iso_l2_k10_d8": dict(
        seed=101, N=300, d=8, k=10, b=32, R=1, kernel="matern15", metric="l2", length_scale=3.0, noise=1e-3,
        probes=[{"length_scale": 1.5}, {"length_scale": 3.0}, {"length_scale": 6.0},
                {"length_scale": 2.0, "noise": 1e-2}],
    ),
    "rbf_iso_F2_k10_d1": dict(
        seed=102, N=400, d=1, k=10, b=25, R=1, kernel="rbf", metric="F2", length_scale=0.7, noise=1e-4,
        probes=[{"length_scale": 0.5}, {"length_scale": 1.1}],
    ),
    "m05_iso_l2_k5_d2": dict(
        seed=103, N=200, d=2, k=5, b=7, R=1, kernel="matern05", metric="l2", length_scale=1.3, noise=1e-5,
    ),
    "m25_iso_l2_k30_d40": dict(
        seed=104, N=500, d=40, k=30, b=16, R=1, kernel="matern25", metric="l2", length_scale=5.0, noise=1e-3,
    ),
    "minf_iso_l2_k16_d3": dict(
        seed=105, N=300, d=3, k=16, b=33, R=1, kernel="maternInf", metric="l2", length_scale=1.5, noise=1e-2,
    ),
    "m15_aniso_l2_k8_d4": dict(
        seed=106, N=250, d=4, k=8, b=12, R=1, kernel="matern15", metric="l2",
        length_scale=[0.8, 1.7, 2.5, 1.1], noise=1e-4,
        probes=[{"length_scale0": 1.0, "length_scale1": 1.0, "length_scale2": 1.0, "length_scale3": 1.0},
                {"length_scale0": 0.5, "length_scale1": 2.0, "length_scale2": 3.0, "length_scale3": 0.9}],
    ),
    "rbf_aniso_F2_k6_d3": dict(
        seed=107, N=200, d=3, k=6, b=9, R=1, kernel="rbf", metric="F2", length_scale=[1.2, 0.6, 2.2], noise=1e-3,
    ),
    "m15_hetero_k10_d5": dict(
        seed=108, N=300, d=5, k=10, b=20, R=1, kernel="matern15", metric="l2", length_scale=2.0, noise=0.0,
        hetero=True,
    ),
    "m15_iso_R3_k12_d6": dict(
        seed=109, N=300, d=6, k=12, b=18, R=3, kernel="matern15", metric="l2", length_scale=2.5, noise=1e-3,
    ),
    "m15_aniso_k50_d8_c4": dict(  # BASELINE config 4 shape
        seed=110, N=600, d=8, k=50, b=8, R=1, kernel="matern15", metric="l2",
        length_scale=[0.7, 1.9, 1.2, 0.6, 1.5, 1.0, 0.9, 1.8], noise=1e-5,
    ),
    "rbf_iso_R16_k64_d40_c5": dict(  # BASELINE config 5 shape
        seed=111, N=400, d=40, k=64, b=4, R=16, kernel="rbf", metric="F2", length_scale=5.0, noise=1e-3,
    ),
    "m15_iso_knn_k30_d40_c2": dict(  # BASELINE config 2 shape, true kNN neighbourhoods
        seed=112, N=2000, d=40, k=30, b=64, R=1, kernel="matern15", metric="l2", length_scale=5.0, noise=1e-3,
        knn=True, probes=[{"length_scale": 4.0}, {"length_scale": 7.5}],
    ),
    "rbf_iso_knn_k10_d1_c1": dict(  # BASELINE config 1 shape (univariate tutorial)
        seed=113, N=1000, d=1, k=10, b=100, R=1, kernel="rbf", metric="F2", length_scale=0.05, noise=1e-5,
        knn=True, spread=0.3,
    ),
    "m15_iso_b1_k3_d2": dict(
        seed=114, N=50, d=2, k=3, b=1, R=1, kernel="matern15", metric="l2", length_scale=1.0, noise=1e-3,
    ),
}


def main():
    for name, c in CASES.items():
        make_case(name, c)
    # sharding rule fixture (_src/mpi_utils.py:36-41)
    rule = {f"{n}_{p}": _get_chunk_sizes(n, p) for n in (0, 1, 7, 64, 257, 1000003) for p in (1, 2, 3, 4, 8)}
    with open(os.path.join(HERE, "chunk_sizes.json"), "w") as f:
        json.dump(rule, f)


if __name__ == "__main__":
    main()
