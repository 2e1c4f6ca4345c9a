"""CPU: backend / ftype selection rules of muygpys_amd.config (reference: _src/config.py:221-261)."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, code):
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_defaults_and_update():
    from muygpys_amd.config import _Config

    c = _Config()
    assert c.state.backend == "hip" and c.muygpys_backend == "hip"
    c.update("muygpys_ftype", "32")
    assert c.state.low_precision()
    with pytest.raises(ValueError):
        c.update("muygpys_backend", "numpy")
    with pytest.raises(ValueError):
        c.update("muygpys_ftype", "16")
    with pytest.raises(AttributeError):
        c.update("muygpys_nope", 1)


def test_env_selects_ftype_and_rejects_unknown_backend():
    r = _run({"MUYGPYS_FTYPE": "32", "MUYGPYS_BACKEND": "hip"},
             "import muygpys_amd._src.math as mm, torch; assert mm.ftype is torch.float32; print('ok')")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
    r = _run({"MUYGPYS_BACKEND": "cuda"}, "import muygpys_amd.config")
    assert r.returncode != 0 and "hip" in r.stderr


def test_device_is_required_not_emulated():
    """No CPU fallback: without a ROCm device the backend refuses, like an unavailable
    reference backend (config.py:230-243), and backend functions reject host arrays."""
    import torch

    from muygpys_amd.config import _Config

    if torch.cuda.is_available():
        pytest.skip("a device is visible")
    with pytest.raises(ValueError):
        _Config().require_device()
    from muygpys_amd._src.gp import tensors as T

    with pytest.raises(TypeError):
        T._F2(torch.zeros((3, 2)))


def test_math_facade_has_no_host_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is visible")
    import muygpys_amd._src.math as mm

    for ctor in (lambda: mm.ones((2, 2)), lambda: mm.array([1.0, 2.0]), lambda: mm.arange(3)):
        with pytest.raises(ValueError):
            ctor()
