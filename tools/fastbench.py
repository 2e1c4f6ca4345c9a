#!/usr/bin/env python3
"""Throughput of the fused fast-posterior-mean prediction kernel (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import random_neighbors, synth
from muygpys_amd.fused import KernelSpec, fast_posterior_mean

n, b, k, d = 1_000_000, 1_000_000, 30, 40
dev = torch.device("cuda")
X, y = synth(n, d, 20241008)
Xd = torch.from_numpy(X).to(dev)
_, ni = random_neighbors(n, b, k, 1)
ni = torch.from_numpy(ni).to(dev)
coeffs = torch.randn((n, k), device=dev)
closest = torch.randint(0, n, (b,), device=dev)
spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
out = torch.empty((b, 1), device=dev)
ts = []
for r in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fast_posterior_mean(spec, Xd, Xd, None, ni, coeffs, closest, out=out)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = float(np.median(ts[2:]))
B = (k + 1) * d * 4 + k * 4 + 8 * (k + 2) + 4
print(f"fast posterior mean: {t:.3f} ms -> {b / t / 1e3:.1f} M test points/s, {B} B/point algorithmic -> "
      f"{B * b / t / 1e6:.0f} GB/s = {B * b / t / 1e6 / 8000:.1%} of 8 TB/s")
