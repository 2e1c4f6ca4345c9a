#!/usr/bin/env python3
"""Generate tests/golden/fp32_reference.npz: what the REAL reference computes in fp32.

The forward fixtures (make_golden.py) hold the reference's fp64 (numpy) results.  How close an fp32
implementation CAN come to them is a property of the arithmetic (``var = Kout - c^T K^-1 c`` is a
difference of O(1) numbers), so the fp32 acceptance band of the HIP kernels is calibrated against the
reference's own fp32 backend instead of a hand-picked floor: this script imports MuyGPyS with
``MUYGPYS_BACKEND=torch MUYGPYS_FTYPE=32`` (README.md:165-177), feeds it the INPUTS
stored in every forward fixture and stores its posterior mean and unscaled variance, per fixture:

    <name>/mean32, <name>/var32        (the reference's torch backend, fp32, CPU)

Run in the build container only (needs /root/reference, which never travels):

    MUYGPYS_BACKEND=torch MUYGPYS_FTYPE=32 PYTHONPATH=/root/reference/src \
        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fp32.py
"""

import glob
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

assert config.state.backend == "torch" and config.state.ftype == "32", "see the module docstring"

from MuyGPyS.gp import MuyGPS  # noqa: E402
from MuyGPyS.gp.deformation import Anisotropy, F2, Isotropy, l2  # noqa: E402
from MuyGPyS.gp.hyperparameter import FixedScale, Parameter, VectorParameter  # noqa: E402
from MuyGPyS.gp.kernels import RBF, Matern  # noqa: E402
from MuyGPyS.gp.noise import HeteroscedasticNoise, HomoscedasticNoise  # noqa: E402
from MuyGPyS.gp.tensors import make_heteroscedastic_tensor  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NU = {"matern05": 0.5, "matern15": 1.5, "matern25": 2.5, "maternInf": float("inf")}


def model(meta, noise_obj):
    metric = l2 if meta["metric"] == "l2" else F2
    ls = meta["length_scale"]
    if isinstance(ls, list):
        deformation = Anisotropy(metric, length_scale=VectorParameter(*[Parameter(float(v)) for v in ls]))
    else:
        deformation = Isotropy(metric, length_scale=Parameter(float(ls)))
    kernel = RBF(deformation=deformation) if meta["kernel"] == "rbf" else Matern(
        smoothness=Parameter(NU[meta["kernel"]]), deformation=deformation)
    return MuyGPS(kernel=kernel, noise=noise_obj, scale=FixedScale())


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if name.startswith(("grad_", "fast_", "gen_", "fp32_", "scale_")):
            continue
        g = np.load(path)
        meta = json.loads(str(g["meta"]))
        X = torch.from_numpy(g["features"]).float()
        y = torch.from_numpy(g["targets"]).float()
        bi, ni = torch.from_numpy(g["batch_idx"]), torch.from_numpy(g["nn_idx"])
        if meta.get("hetero"):
            noise_obj = HeteroscedasticNoise(make_heteroscedastic_tensor(torch.from_numpy(g["noise_table"]).float(), ni))
        else:
            noise_obj = HomoscedasticNoise(meta["noise"])
        m = model(meta, noise_obj)
        cross, pair, y_b, y_nn = m.make_train_tensors(bi, ni, X, y)
        Kin, Kc = m.kernel(pair), m.kernel(cross)
        assert Kin.dtype == torch.float32, Kin.dtype
        mean = m.posterior_mean(Kin, Kc, y_nn)
        var = m.get_opt_var_fn()(Kin, Kc)
        out[name + "/mean32"] = mean.detach().numpy().astype(np.float32)
        out[name + "/var32"] = var.detach().numpy().astype(np.float32)
        e_m = np.abs(out[name + "/mean32"].reshape(g["mean"].shape) - g["mean"]).max()
        e_v = (np.abs(out[name + "/var32"] - g["var_unscaled"]) / np.abs(g["var_unscaled"])).max()
        print(f"{name:28s} reference fp32 vs fp64: max |mean err| {e_m:.2e}   max rel var err {e_v:.2e}")
    np.savez_compressed(os.path.join(HERE, "fp32_reference.npz"), **out)


if __name__ == "__main__":
    main()
