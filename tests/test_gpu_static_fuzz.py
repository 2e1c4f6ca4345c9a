"""GPU: randomized parity of the statically specialised fp32 kernels (run-time compiled, folded elimination where it
applies) against the fp64 LDS workgroup kernel, over shapes, kernels, Isotropy / Anisotropy, homo- / hetero-scedastic
noise, prepared / plain tables and batch sizes from 1 up.  Runs in a child process with MUYGPYS_HIP_JIT=force so that
even one-neighbourhood calls take the specialised kernel (the policy is read once per process)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var
assert _lib.load().mgp_jit_mode() == 2
rng = np.random.default_rng(%(seed)d)
gen = torch.Generator(device="cuda").manual_seed(%(seed)d)
rows = []
for trial in range(%(trials)d):
    k = int(rng.integers(8, 62)); R = int(rng.integers(1, 5))
    if k + 1 + R > 64: R = 1
    d = int(rng.choice([4, 8, 12, 16, 24, 32, 40, 48, 56, 64]))
    kern = str(rng.choice(["matern15", "matern25", "rbf", "maternInf", "matern05"]))
    metric = "F2" if kern == "rbf" else "l2"
    aniso, hetero = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    packed = bool(rng.integers(0, 2))
    b = int(rng.choice([1, 2, 3, 5, 257, 1000, 4099]))
    N = 3000
    X = torch.randn(N, d, device="cuda", generator=gen); y = torch.randn(N, R, device="cuda", generator=gen)
    bi = torch.randint(0, N, (b,), device="cuda", generator=gen); ni = torch.randint(0, N, (b, k), device="cuda", generator=gen)
    ls = (np.sqrt(2 * d) * rng.uniform(0.7, 1.4, size=d)).tolist() if aniso else float(np.sqrt(2 * d))
    noise = (10.0 ** torch.empty(N, device="cuda").uniform_(-3, -2, generator=gen)) if hetero else 1e-3
    def run(dt, path, pk):
        nz = noise.to(dt) if hetero else noise
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = posterior_mean_var(KernelSpec(kern, metric, ls, nz), X.to(dt), X.to(dt), bi, ni, y.to(dt), want_ykinvy=True,
                                 path=path, packed=pk, info=info)
        return out, int(info.item())
    before = _lib.load().mgp_jit_loaded_count()
    (got, bad), (truth, _) = run(torch.float32, "auto", packed), run(torch.float64, "generic", False)
    torch.cuda.synchronize()
    err = max(float((g.double().reshape(-1) - t.reshape(-1)).abs().max() / t.abs().max()) for g, t in zip(got, truth))
    rows.append(dict(k=k, R=R, d=d, kernel=kern, aniso=aniso, hetero=hetero, packed=packed, b=b, err=err, bad=bad,
                     specialised=_lib.served_by(d, k, R, torch.float32, packed, "auto")))
print(json.dumps(rows))
"""


@pytest.mark.parametrize("seed", [11, 12])
def test_specialised_kernels_match_fp64_over_random_configurations(seed):
    env = dict(os.environ, MUYGPYS_HIP_JIT="force", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "seed": seed, "trials": 24}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(rows) == 24
    static = [x for x in rows if "0,0,0" not in x["specialised"].replace(" ", "")]
    assert len(static) >= 12, "most configurations must be served by a specialised kernel"
    for x in rows:
        assert x["bad"] == 0, x
        assert x["err"] < 1e-3, x  # fp32 against fp64, relative to the largest magnitude of the output
