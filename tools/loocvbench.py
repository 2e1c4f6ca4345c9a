#!/usr/bin/env python3
"""LOOCV objective evaluations per second (config 3 shape, one GPU): fused launch with
y^T K^-1 y + two fp64 reductions + host finish, as the optimiser drivers call it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import random_neighbors, synth
from muygpys_amd import distributed as D
from muygpys_amd.fused import KernelSpec

n, b, k, d = 1_000_000, 1_000_000, 30, 40
dev = torch.device("cuda")
X, y = synth(n, d, 20241008)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
for ls in (3.0, 5.0, 8.0):
    res = D.sharded_loocv(KernelSpec("matern15", "l2", ls, 1e-3), Xd, yd, bi, ni)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for r in range(R):
    res = D.sharded_loocv(KernelSpec("matern15", "l2", 4.0 + 0.1 * r, 1e-3), Xd, yd, bi, ni)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f"LOOCV objective: {dt * 1e3:.3f} ms per evaluation over {b} neighbourhoods -> {b / dt / 1e6:.1f} M nbhd/s; "
      f"lool={res['lool']:.6g} sigma_sq={res['sigma_sq']:.6g}")
