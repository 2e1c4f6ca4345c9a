import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _process_wide_state_is_put_back():
    """config.state survives a test (the library is one process-wide backend): a test that changes the
    not-positive-definite check mode or the lazy-tensor switch must not decide what the tests after it see."""
    from muygpys_amd.config import config as mconfig

    spd, lazy = mconfig.state.check_spd, mconfig.state.lazy_tensors
    yield
    mconfig.state.check_spd, mconfig.state.lazy_tensors = spd, lazy
    from muygpys_amd import _lib

    _lib._SPD_PENDING.clear()


def _npz_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """Forward-path fixtures (tests/golden/make_golden.py)."""
    return [n for n in _npz_names() if not n.startswith(("grad_", "fast_", "gen_", "fp32_", "scale_"))]


def fast_golden_names():
    """Fast-posterior-mean fixtures from the reference's own workflow (make_golden_fast.py)."""
    return [n for n in _npz_names() if n.startswith("fast_")]


def grad_golden_names():
    """Gradient fixtures from the reference's torch backend + autograd (make_golden_grad.py)."""
    return [n for n in _npz_names() if n.startswith("grad_")]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta"]))
    return g


def spec_from_meta(meta, g):
    """Build an oracle Spec from a fixture's meta block."""
    from oracle.muygps_oracle import Spec

    ls = meta["length_scale"]
    ls = np.asarray(ls, dtype=np.float64) if isinstance(ls, list) else float(ls)
    noise = g["noise_table"] if meta.get("hetero") else float(meta["noise"])
    return Spec(kernel=meta["kernel"], metric=meta["metric"], length_scale=ls, noise=noise)


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)


def grad_spec(meta, g):
    """Oracle Spec + query table of a gradient fixture."""
    from oracle.muygps_oracle import Spec

    ls = np.asarray(meta["ls"], dtype=np.float64) if isinstance(meta["ls"], list) else float(meta["ls"])
    noise = g["noise_table"] if meta["hetero"] else float(meta["eps"])
    xq = g["test_features"] if meta["separate_test"] else g["features"]
    return Spec(kernel=meta["kernel"], metric=meta["metric"], length_scale=ls, noise=noise), xq


@pytest.fixture(params=grad_golden_names())
def grad_golden(request):
    return load_golden(request.param)
