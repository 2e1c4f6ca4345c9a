"""Backend / float-width selection, mirroring the slice of MuyGPyS's config the path needs.

Reference: ``MUYGPYS_BACKEND`` and ``MUYGPYS_FTYPE`` are read once at import
(src/MuyGPyS/_src/config.py:221-261, jaxconfig.py:323); ``config.update(name, value)`` may
change them before the backend modules are imported (README.md:139-143); the current values
are mirrored in ``config.state`` (config.py:42-54).  This package provides exactly one
backend, ``hip``; the reference's own values (numpy, jax, torch, mpi) are rejected with the
same ``ValueError`` style as an unknown backend there (config.py:239-243).
"""

from __future__ import annotations

import os

_BACKENDS = ("hip",)
_REFERENCE_BACKENDS = ("numpy", "jax", "torch", "mpi")
_FTYPES = ("64", "32")


class _State:
    def __init__(self):
        self.backend = "hip"
        self.ftype = "64"
        self.hip_enabled = None  # resolved lazily: a visible ROCm device
        # raise numpy.linalg.LinAlgError (like the reference's linalg.solve on a singular system)
        # when a neighbourhood is not positive definite; costs one device sync per solve call,
        # MUYGPYS_HIP_CHECK_SPD=0 turns it off (outputs of such neighbourhoods are then NaN);
        # "deferred" (MUYGPYS_HIP_CHECK_SPD=deferred): the counter goes to pinned host memory behind the launch and is
        # read when the NEXT checked call arrives or at _lib.flush_spd_checks() -- the error is raised one call late,
        # and the host no longer waits for every launch (a loop of evaluations runs 5 % faster: bench.py --route dropin)
        _spd = os.environ.get("MUYGPYS_HIP_CHECK_SPD", "1")
        self.check_spd = "deferred" if _spd == "deferred" else _spd != "0"
        # the tensor family returns lazy handles (muygpys_amd.lazy) instead of (b, k, k, d) tensors,
        # so that a functor layer written against the family functions -- the reference's own, after
        # integration.install() -- reaches the fused launch; off by default: called directly, the
        # family functions are the materialising per-function kernels
        self.lazy_tensors = os.environ.get("MUYGPYS_HIP_LAZY", "0") == "1"

    def low_precision(self) -> bool:
        """config.state.low_precision(), config.py:53."""
        return self.ftype == "32"


class _Config:
    def __init__(self):
        self.state = _State()
        env_backend = os.environ.get("MUYGPYS_BACKEND", "hip")
        # numpy/jax/torch/mpi select one of the REFERENCE package's backends (this package may be
        # loaded next to it, see integration.py); they are not a request addressed to this package
        self.update("muygpys_backend", "hip" if env_backend in _REFERENCE_BACKENDS else env_backend)
        self.update("muygpys_ftype", os.environ.get("MUYGPYS_FTYPE", "64"))

    def update(self, name: str, value) -> None:
        value = str(value)
        if name == "muygpys_backend":
            if value not in _BACKENDS:
                raise ValueError(
                    f'muygpys_amd provides the backend "hip" only; got MUYGPYS_BACKEND="{value}" '
                    "(numpy/jax/torch/mpi are the reference package's backends)"
                )
            self.state.backend = value
        elif name == "muygpys_ftype":
            if value not in _FTYPES:
                raise ValueError(f'MUYGPYS_FTYPE must be one of {_FTYPES}, got "{value}"')
            self.state.ftype = value
        else:
            raise AttributeError(f"Unrecognized config option: {name}")

    @property
    def muygpys_backend(self) -> str:
        return self.state.backend

    @property
    def muygpys_ftype(self) -> str:
        return self.state.ftype

    def require_device(self) -> None:
        """The hip backend needs a visible ROCm device (reference: a backend whose runtime is
        absent is refused with ValueError, config.py:230-243)."""
        if self.state.hip_enabled is None:
            import torch

            self.state.hip_enabled = bool(torch.cuda.is_available())
        if not self.state.hip_enabled:
            raise ValueError('MuyGPyS backend "hip" needs a visible ROCm device; none was found.')


config = _Config()
