#!/usr/bin/env python3
"""Generate tests/golden/grad_*.npz: gradients of the hot path from the REAL reference.

The reference has no hand-written backward pass: a deep-kernel model trains through
torch autograd over the torch backend (torch/muygps_layer.py:129-164,
examples/muygps_torch.py:425-437).  This script imports MuyGPyS with
``MUYGPYS_BACKEND=torch MUYGPYS_FTYPE=64``, runs exactly that op sequence
(deformation.crosswise_tensor / pairwise_tensor -> kernel -> posterior mean / variance)
on seeded inputs with ``requires_grad`` leaves, contracts the outputs with seeded cotangents
and stores inputs + outputs + every leaf gradient as data.

Run in the build container only (needs /root/reference, which never travels):

    MUYGPYS_BACKEND=torch MUYGPYS_FTYPE=64 PYTHONPATH=/root/reference/src \
        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad.py
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
import torch  # noqa: E402

from MuyGPyS import config  # noqa: E402

assert config.state.backend == "torch" and config.state.ftype == "64", "see the module docstring"

from MuyGPyS.gp import MuyGPS  # noqa: E402
from MuyGPyS.gp.deformation import Anisotropy, F2, Isotropy, l2  # noqa: E402
from MuyGPyS.gp.hyperparameter import FixedScale, Parameter, VectorParameter  # noqa: E402
from MuyGPyS.gp.kernels import RBF, Matern  # noqa: E402
from MuyGPyS.gp.noise import HeteroscedasticNoise, HomoscedasticNoise  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NU = {"matern05": 0.5, "matern15": 1.5, "matern25": 2.5, "maternInf": float("inf")}

CASES = [
    # name, kernel, metric, n, d, k, b, R, length_scale, noise, extras
    dict(name="grad_m15_iso_l2_k6_d3_R2", kernel="matern15", metric="l2", n=60, d=3, k=6, b=9, R=2, ls=1.3, eps=1e-3),
    dict(name="grad_m05_iso_l2_k5_d2", kernel="matern05", metric="l2", n=40, d=2, k=5, b=8, R=1, ls=0.7, eps=1e-2,
         duplicate_rows=True),
    dict(name="grad_rbf_iso_F2_k8_d4", kernel="rbf", metric="F2", n=80, d=4, k=8, b=10, R=1, ls=2.0, eps=1e-3),
    dict(name="grad_m25_aniso_l2_k7_d5_R3", kernel="matern25", metric="l2", n=70, d=5, k=7, b=9, R=3,
         ls=[0.6, 1.1, 1.9, 0.8, 1.4], eps=1e-4),
    dict(name="grad_minf_iso_l2_k10_d1", kernel="maternInf", metric="l2", n=90, d=1, k=10, b=12, R=1, ls=0.9, eps=1e-3),
    dict(name="grad_rbf_aniso_F2_k6_d3", kernel="rbf", metric="F2", n=50, d=3, k=6, b=7, R=1, ls=[1.5, 0.8, 2.2], eps=1e-3),
    dict(name="grad_m15_hetero_k8_d4", kernel="matern15", metric="l2", n=60, d=4, k=8, b=9, R=1, ls=1.1, eps=None,
         hetero=True),
    dict(name="grad_m15_iso_predict_k9_d6_R2", kernel="matern15", metric="l2", n=80, d=6, k=9, b=11, R=2, ls=1.7,
         eps=1e-3, separate_test=True),
    dict(name="grad_m15_iso_k30_d40", kernel="matern15", metric="l2", n=400, d=40, k=30, b=48, R=1, ls=5.0, eps=1e-3,
         knn=True),
    dict(name="grad_m25_iso_l2_k12_d3", kernel="matern25", metric="l2", n=70, d=3, k=12, b=9, R=1, ls=1.6, eps=1e-3),
]


def build(c, noise_obj):
    metric = l2 if c["metric"] == "l2" else F2
    if isinstance(c["ls"], list):
        deformation = Anisotropy(
            metric, length_scale=VectorParameter(*[Parameter(float(v), (0.05, 20.0)) for v in c["ls"]])
        )
    else:
        deformation = Isotropy(metric, length_scale=Parameter(float(c["ls"]), (0.05, 20.0)))
    if c["kernel"] == "rbf":
        kernel = RBF(deformation=deformation)
    else:
        kernel = Matern(smoothness=Parameter(NU[c["kernel"]]), deformation=deformation)
    return MuyGPS(kernel=kernel, noise=noise_obj, scale=FixedScale())


def run(c, seed):
    rng = np.random.default_rng(seed)
    n, d, k, b, R = c["n"], c["d"], c["k"], c["b"], c["R"]
    X = rng.normal(size=(n, d))
    if c.get("duplicate_rows"):
        X[1] = X[0]  # a zero off-diagonal distance: torch.norm's subgradient there is 0
    W = rng.normal(size=(d, R)) / np.sqrt(d)
    Y = np.sin(X @ W) + 0.1 * rng.normal(size=(n, R))
    if c.get("knn"):
        from sklearn.neighbors import NearestNeighbors

        bi = rng.choice(n, size=b, replace=False)
        nn = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(X).kneighbors(X[bi], return_distance=False)[:, 1:]
    else:
        bi = rng.choice(n, size=b, replace=False)
        nn = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
        if c.get("duplicate_rows"):
            nn[0, :2] = [0, 1]
    Xq = rng.normal(size=(b + 3, d)) if c.get("separate_test") else None
    if Xq is not None:
        bi = rng.permutation(b + 3)[:b]
    gm = rng.normal(size=(b, R))
    gv = rng.normal(size=(b,))

    x = torch.tensor(X, requires_grad=True)
    y = torch.tensor(Y, requires_grad=True)
    xq = torch.tensor(Xq, requires_grad=True) if Xq is not None else None
    bi_t, nn_t = torch.tensor(bi), torch.tensor(nn)
    out = dict(features=X, targets=Y, batch_indices=bi, nn_indices=nn, grad_mean=gm, grad_var=gv)
    if Xq is not None:
        out["test_features"] = Xq

    aniso = isinstance(c["ls"], list)
    ls_leaf = torch.tensor(np.asarray(c["ls"], dtype=np.float64), requires_grad=True)

    if c.get("hetero"):
        noise_table = 10.0 ** rng.uniform(-4, -1, size=n)
        nz_table = torch.tensor(noise_table, requires_grad=True)
        nz = nz_table[nn_t]
        model = build(c, HeteroscedasticNoise(nz))
        noise_kwargs = {}
        out["noise_table"] = noise_table
    else:
        nz = torch.tensor(float(c["eps"]), requires_grad=True)
        model = build(c, HomoscedasticNoise(float(c["eps"]), (1e-8, 1.0)))
        noise_kwargs = {"noise": nz}

    deformation = model.kernel.deformation
    crosswise = deformation.crosswise_tensor(x if xq is None else xq, x, bi_t, nn_t)
    pairwise = deformation.pairwise_tensor(x, nn_t)
    if aniso:
        # Anisotropy.__call__ (gp/deformation/anisotropy.py:70) assembles its length-scale vector
        # from python scalars, which detaches it; to keep the leaf in the graph the same expression
        # -- metric(diffs / length_scale) -- is evaluated with the tensor, then the model's own
        # kernel function is applied.
        kfn = model.kernel._predef_fn if hasattr(model.kernel, "_predef_fn") else model.kernel._kernel_fn
        Kcross = kfn(deformation.metric(crosswise / ls_leaf))
        Kin = kfn(deformation.metric(pairwise / ls_leaf))
    else:
        Kcross = model.kernel(crosswise, length_scale=ls_leaf)
        Kin = model.kernel(pairwise, length_scale=ls_leaf)
    mean = model.get_opt_mean_fn()(Kin, Kcross, y[nn_t], **noise_kwargs)
    var = model.get_opt_var_fn()(Kin, Kcross, **noise_kwargs)
    assert mean.shape == (b, R) and var.shape == (b,), (mean.shape, var.shape)
    ((mean * torch.tensor(gm)).sum() + (var * torch.tensor(gv)).sum()).backward()

    out.update(
        mean=mean.detach().numpy(),
        var=var.detach().numpy(),
        g_features=x.grad.numpy(),
        g_targets=y.grad.numpy(),
        g_length_scale=np.atleast_1d(ls_leaf.grad.numpy()),
    )
    if xq is not None:
        out["g_test_features"] = xq.grad.numpy()
    if c.get("hetero"):
        out["g_noise_table"] = nz_table.grad.numpy()
    else:
        out["g_noise"] = np.array(nz.grad.item())
    meta = {key: c[key] for key in ("kernel", "metric", "ls", "eps", "R", "k", "d")}
    meta["hetero"] = bool(c.get("hetero"))
    meta["separate_test"] = bool(c.get("separate_test"))
    out["meta"] = np.array(json.dumps(meta))
    return out


if __name__ == "__main__":
    for i, c in enumerate(CASES):
        data = run(c, 7000 + i)
        path = os.path.join(HERE, c["name"] + ".npz")
        np.savez_compressed(path, **data)
        print(f"{c['name']}: {os.path.getsize(path)} B, |g_x|={np.abs(data['g_features']).sum():.4g} "
              f"g_ls={data['g_length_scale']}")
