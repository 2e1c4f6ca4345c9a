#!/usr/bin/env python3
"""General-smoothness Matern on the fused path, fp32 and fp64, against the fixed nu = 3/2 closed form (GPU box):
python tools/genbench.py  -> one JSON line (ms per launch, M neighbourhoods/s, kernel that served the call)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var


def main():
    out = {}
    gen = torch.Generator(device="cuda").manual_seed(5)
    for dtype, (N, d, k, b) in (("f32", (400_000, 40, 30, 400_000)), ("f64", (400_000, 40, 30, 200_000)),
                                ("f64_k10_d6", (300_000, 6, 10, 300_000))):
        td = torch.float32 if dtype == "f32" else torch.float64
        X = torch.randn(N, d, device="cuda", generator=gen).to(td)
        y = torch.randn(N, device="cuda", generator=gen).to(td)
        bi = torch.arange(b, device="cuda")
        ni = torch.randint(0, N - 1, (b, k), device="cuda", generator=gen)
        ni = ni + (ni >= bi[:, None])
        ell = float(np.sqrt(d / 40.0) * 6.0)
        res = {}
        for name, spec in (("fixed_nu_1.5", KernelSpec("matern15", "l2", ell, 1e-3)),
                           ("free_nu_1.5", KernelSpec("matern_gen", "l2", ell, 1e-3, smoothness=1.5)),
                           ("free_nu_0.8", KernelSpec("matern_gen", "l2", ell, 1e-3, smoothness=0.8)),
                           ("free_nu_4.2", KernelSpec("matern_gen", "l2", ell, 1e-3, smoothness=4.2))):
            ts = []
            for r in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                m, v = posterior_mean_var(spec, X, X, bi, ni, y, packed=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts[2:]))
            res[name] = {"ms": ms, "M_nbhd_per_s": b / ms / 1e3, "kernel": _lib.last_kernel(), "finite": bool(torch.isfinite(m).all())}
        out[dtype] = {"points": N, "d": d, "k": k, "batch": b, **res}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
