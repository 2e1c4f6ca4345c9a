"""Install the hip backend into an importable LLNL/MuyGPyS (the reference package itself).

MuyGPyS resolves its backend once, at import of each family module, through
``MuyGPyS._src.util._collect_implementation(package, *names)`` -> ``import package.<backend>``
(src/MuyGPyS/_src/util.py:9-32), and accepts only numpy/jax/torch/mpi
(_src/config.py:223).  ``install()`` makes ``hip`` a fifth backend WITHOUT editing the
reference tree:

    import MuyGPyS                      # config only; no family module is bound yet
    import muygpys_amd.integration as hip_backend
    hip_backend.install()               # before the first `import MuyGPyS.gp ...`
    from MuyGPyS.gp import MuyGPS       # now backed by the HIP kernels

What a maintainer would commit instead is shown in INTEGRATION.md: eight three-line
``hip.py`` files and one ``elif`` in util.py.

With ``lazy=True`` (the default) the tensor family returns the light handles of
``muygpys_amd.lazy`` instead of (b, k, k, d) difference tensors and the kernel / noise / solve /
scale families decorate and finally evaluate them, so the reference's OWN functor layer --
``MuyGPS.make_train_tensors -> kernel -> posterior_mean / posterior_variance / optimize_scale``,
``make_loo_crossval_fn`` -- ends in one fused launch per evaluation (``mgp_posterior_gathered_*``:
the reference gathers the neighbour responses itself).  ``lazy=False`` binds the materialising
per-function kernels.
"""

from __future__ import annotations

import importlib
import sys

FAMILIES = (
    "math", "gp.tensors", "gp.kernels", "gp.muygps", "gp.noise", "optimize.loss", "optimize.scale", "optimize.chassis",
)


def install(package: str = "MuyGPyS", require_device: bool = True, lazy: bool = True) -> None:
    ref = importlib.import_module(package)
    bound = [m for m in sys.modules if m.startswith(f"{package}._src.") and m.split(".")[-1] in
             ("tensors", "kernels", "muygps", "noise", "loss", "scale", "chassis")]
    if bound:
        raise RuntimeError(
            f"install() must run before the backend families are imported; already bound: {sorted(bound)}"
        )
    from muygpys_amd.config import config as hip_config

    if require_device:
        hip_config.require_device()
    hip_config.state.lazy_tensors = bool(lazy)
    # 1. the family modules MuyGPyS will look for
    for fam in FAMILIES:
        sys.modules[f"{package}._src.{fam}.hip"] = importlib.import_module(f"muygpys_amd._src.{fam}.hip")
    # 2. the resolver learns the fifth name
    util = importlib.import_module(f"{package}._src.util")
    original = util._collect_implementation

    def _collect_implementation(pkg, *funcs):
        if ref.config.state.backend == "hip":
            return util._collect_functions(pkg + ".hip", *funcs)
        return original(pkg, *funcs)

    util._collect_implementation = _collect_implementation
    # 3. select it (the enum validator only knows the four stock names, so set the state mirror)
    ref.config.state.backend = "hip"
    ref.config.state.ftype = hip_config.state.ftype


def table(t):
    """Wrap a device tensor (no copy) so that ``t[batch_nn_indices]`` -- the reference's own response
    gather, gp/muygps.py:474 -- stays a lazy handle and the fused launch reads the responses from the
    prepared table.  Tensors made by the facade's constructors (``mm.array`` ...) already are."""
    import torch

    from muygpys_amd._src.math.hip import TableTensor

    return t if isinstance(t, TableTensor) else torch.as_tensor(t).as_subclass(TableTensor)
