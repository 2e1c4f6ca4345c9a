#!/usr/bin/env python3
"""A/B timing of one LOOCV evaluation (mgp_loocv_packed_f32, headline shape) over library variants
(tools/mkvariant.sh; MUYGPYS_HIP_LIB), each in its own process, rounds interleaved:
    python tools/loocv_ab.py --variants cur,noarrive,... [--b 125000] [--rounds 3]
`cur` = the built library.  Prints per variant: back-to-back HIP-event time per evaluation (median over rounds) and the
prediction kernel (mgp_posterior_packed_f32) beside it."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from bench import random_neighbors, synth
from muygpys_amd.fused import KernelSpec, posterior_mean_var, loocv_partials
b = int(sys.argv[1]); n = 1_000_000; k, d = 30, 40
X, y = synth(n, d, 20241008, 1)
Xd, yd = torch.from_numpy(X).cuda().float(), torch.from_numpy(y).cuda().float()
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).cuda(), torch.from_numpy(ni).cuda()
spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
mean = torch.empty((b, 1), device="cuda"); var = torch.empty((b,), device="cuda")
def ev(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
reps = max(40, int(60e-3 / (b * 1.6e-9)))
out = {"pred": ev(lambda: posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, packed=True), reps, reps // 2)}
try:
    out["loocv"] = ev(lambda: loocv_partials(spec, Xd, yd, bi, ni, packed=True), reps, reps // 2)
    p = loocv_partials(spec, Xd, yd, bi, ni, packed=True)[0]; torch.cuda.synchronize()
    out["sum0"] = float(p[0].item())
except Exception as e:
    out["loocv"] = None
print(json.dumps(out))
"""
ap = argparse.ArgumentParser()
ap.add_argument("--variants", required=True); ap.add_argument("--b", type=int, default=125_000); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
names = a.variants.split(",")
res = {v: [] for v in names}
for r in range(a.rounds):
    for v in names:
        env = dict(os.environ)
        if v != "cur": env["MUYGPYS_HIP_LIB"] = os.path.join(ROOT, "variants", f"lib_{v}.so")
        o = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(a.b)], env=env, capture_output=True, text=True)
        try: res[v].append(json.loads(o.stdout.strip().splitlines()[-1]))
        except Exception: print(v, "failed:", o.stderr[-400:])
import statistics
for v in names:
    if not res[v]: continue
    med = lambda key: statistics.median([x[key] for x in res[v] if x.get(key) is not None]) if any(x.get(key) is not None for x in res[v]) else float("nan")
    print(f"{v:10s} b={a.b}: prediction {med('pred'):.4f} ms   loocv {med('loocv'):.4f} ms   sum0 {res[v][0].get('sum0')}")
