#!/usr/bin/env python3
"""Where one LOOCV objective evaluation spends its time (GPU box): python tools/c3bench.py [--b 125000]
kernel-only (back-to-back launches, HIP events), the library call with its partial sums, the host cost of issuing it,
and the whole evaluation with the host read-back (what an optimiser sees)."""
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synth, random_neighbors
from muygpys_amd import distributed as D
from muygpys_amd.fused import KernelSpec, posterior_mean_var, loocv_partials
ap = argparse.ArgumentParser(); ap.add_argument("--b", type=int, default=1_000_000); ap.add_argument("--n", type=int, default=1_000_000)
a = ap.parse_args()
n, b, k, d = a.n, a.b, 30, 40
X, y = synth(n, d, 1, 1)
Xd, yd = torch.from_numpy(X).cuda().float(), torch.from_numpy(y).cuda().float()
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).cuda(), torch.from_numpy(ni).cuda()
spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
def ev(fn, reps=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def wall(fn, reps=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def host(fn, reps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return dt
mean = torch.empty((b, 1), device="cuda"); var = torch.empty((b,), device="cuda")
print(f"b = {b}")
print("kernel, back to back (events)      %.4f ms" % ev(lambda: posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var)))
print("loocv_partials, back to back        %.4f ms" % ev(lambda: loocv_partials(spec, Xd, yd, bi, ni)))
print("loocv_partials, host issue cost     %.4f ms" % host(lambda: loocv_partials(spec, Xd, yd, bi, ni)))
print("loocv + tolist (sync per step)      %.4f ms" % wall(lambda: loocv_partials(spec, Xd, yd, bi, ni)[0].tolist()))
print("sharded_loocv (bench step)          %.4f ms" % wall(lambda: D.sharded_loocv(spec, Xd, yd, bi, ni, presharded=True)))
