"""CPU, build container only: the hip family modules install into the real MuyGPyS package and
every name the reference's family __init__ binds resolves to a hip function."""

import os
import subprocess
import sys

import pytest

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib.metadata as md, sys, types
_v = md.version; md.version = lambda n: "0.9.0" if n == "MuyGPyS" else _v(n)
bo = types.ModuleType("bayes_opt"); bo.BayesianOptimization = object; sys.modules["bayes_opt"] = bo
import MuyGPyS
import muygpys_amd.integration as hip_backend
hip_backend.install(require_device=False)
import MuyGPyS._src.math as mm
import MuyGPyS._src.gp.tensors as T, MuyGPyS._src.gp.kernels as K, MuyGPyS._src.gp.muygps as M
import MuyGPyS._src.gp.noise as N, MuyGPyS._src.optimize.loss as L, MuyGPyS._src.optimize.scale as S
import MuyGPyS._src.optimize.chassis as C
for mod, names in ((T, ["_crosswise_tensor", "_pairwise_tensor", "_F2", "_l2"]), (K, ["_rbf_fn", "_matern_15_fn"]),
                   (M, ["_muygps_posterior_mean", "_muygps_diagonal_variance"]), (N, ["_homoscedastic_perturb"]),
                   (L, ["_lool_fn", "_mse_fn"]), (S, ["_analytic_scale_optim"]), (C, ["_scipy_optimize"])):
    for n in names:
        assert getattr(mod, n).__module__.startswith("muygpys_amd._src."), (mod.__name__, n)
import torch
assert mm.ndarray is torch.Tensor and MuyGPyS.config.state.backend == "hip"
from MuyGPyS.gp import MuyGPS          # the reference's functor layer now sits on the hip functions
from MuyGPyS.gp.kernels import Matern
print("installed")
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")
def test_install_into_reference_package():
    env = dict(os.environ, PYTHONPATH=REF + os.pathsep + ROOT, PYTHONDONTWRITEBYTECODE="1", MUYGPYS_BACKEND="numpy")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "installed" in r.stdout, r.stderr[-2000:]
