#!/usr/bin/env python3
"""Phase timing of the rhs-column kernel's FOLDED variant (a library built with -DMGP_RHS_TIMING=1 -DMGP_RHS_MF=0:
tools/mkvariant.sh timing mgp_fused_rhs.hip -DMGP_RHS_TIMING=1 -DMGP_RHS_MF=0 -- with MGP_RHS_MF the config-5 shape goes
to mgp_fused_rhs_mf.hip, which carries no stamps).  Prints the share of a wave's life spent per phase on BASELINE config 5.

    MUYGPYS_HIP_LIB=variants/lib_timing.so python tools/rhs_timing.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import random_neighbors, synth
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var

n, b, k, d, R = 2_000_000, 500_000, 64, 40, 16
dev = torch.device("cuda")
X, y = synth(n, d, 20241008, R)
Xd, yd = torch.from_numpy(X).to(dev, torch.float32), torch.from_numpy(y).to(dev, torch.float32)
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
spec = KernelSpec("rbf", "l2", 5.0, 1e-3)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
mf = "--mf" in sys.argv  # the matrix-core-layout kernel (a library built with -DMGP_RHS_MF_TIMING=1 on mgp_fused_rhs_mf.hip)
read = lib.mgp_debug_rhs_mf_timing if mf else lib.mgp_debug_rhs_timing
for it in range(3):
    posterior_mean_var(spec, Xd, Xd, bi, ni, yd, packed=False)
    torch.cuda.synchronize()
    read(out, 1)
print(_lib.last_kernel())
names = (["indices + gather", "centring + Gram (MFMA)", "distances + covariances", "elimination", "back-substitution", "outputs", "-", "-"]
         if mf else ["gather", "distances", "cov+exchange+readback", "elimination", "dump/redistribute", "back-substitution", "outputs", "-"])
tot = sum(out)
for nm, v in zip(names, out):
    print(f"{nm:24s} {v / tot * 100:6.1f} %   {v / b:10.1f} ticks per neighbourhood")
