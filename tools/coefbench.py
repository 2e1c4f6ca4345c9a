#!/usr/bin/env python3
"""Timing of the fast-posterior-mean coefficient precompute (GPU box only): fused launch (fp32,
one response, k <= 62) vs the materialising per-function path."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muygpys_amd.fused import KernelSpec, fast_coefficients
from muygpys_amd.neighbors import NN_Wrapper
torch.manual_seed(0)
n, d, k = 1_000_000, 40, 30
X = torch.randn(n, d, device='cuda'); y = torch.randn(n, device='cuda')
nn = NN_Wrapper(X, k).get_batch_nns(torch.arange(n, device='cuda'))[0]
spec = KernelSpec('matern15', 'l2', 5.0, 1e-3)
for label, Xv, yv, fused in (("fused fp32", X, y, True), ("fused fp64", X.double(), y.double(), True),
                             ("per-function fp32", X, y, False), ("per-function fp64", X.double(), y.double(), False)):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, nnf = fast_coefficients(spec, Xv, yv, nn, fused=fused)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fast_coefficients, {n} points, k={k}, d={d}, {label}: {dt * 1e3:.1f} ms -> {n / dt / 1e6:.1f} M points/s")
