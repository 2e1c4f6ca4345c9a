#!/usr/bin/env python3
"""A/B timing of library variants (tools/mkvariant.sh) on one shape: each variant runs in its own
process (one dlopen per process), several rounds interleaved, HIP-event time of the fused launch.

    python tools/abtime.py --variants v0,vA,vB [--k 30 --d 40 --b 1000000 --rounds 3 --dtype f32]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from bench import random_neighbors, synth
from muygpys_amd.fused import KernelSpec, posterior_mean_var
a = json.loads(sys.argv[1])
dev = torch.device("cuda"); td = torch.float32 if a["dtype"] == "f32" else torch.float64
X, y = synth(a["n"], a["d"], 20241008, a["R"])
Xd, yd = torch.from_numpy(X).to(dev, td), torch.from_numpy(y).to(dev, td)
bi, ni = random_neighbors(a["n"], a["b"], a["k"], 1)
bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
ls = [5.0] * a["d"] if a["aniso"] else 5.0
spec = KernelSpec(a["kernel"], "l2", ls, 1e-3)
mean = torch.empty((a["b"], a["R"]), device=dev, dtype=td); var = torch.empty((a["b"],), device=dev, dtype=td)
ts = []
for r in range(a["iters"] + 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, packed=bool(a["packed"])); e1.record()
    torch.cuda.synchronize()
    if r >= 3: ts.append(e0.elapsed_time(e1))
print(json.dumps({"median": float(np.median(ts)), "min": float(min(ts)), "csum": float(mean.double().sum().item()), "vsum": float(var.double().sum().item())}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--b", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--R", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--kernel", default="matern15")
    ap.add_argument("--aniso", type=int, default=0)
    ap.add_argument("--packed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    names = args.variants.split(",")
    res = {v: [] for v in names}
    sums = {}
    for r in range(args.rounds):
        for v in names:
            env = dict(os.environ)
            if v != "default":
                env["MUYGPYS_HIP_LIB"] = os.path.join(ROOT, "variants", f"lib_{v}.so")
            out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, json.dumps(vars(args))], env=env,
                                 capture_output=True, text=True)
            if out.returncode != 0:
                print(v, "FAILED", out.stderr[-400:])
                continue
            for ln in sorted({ln for ln in out.stderr.splitlines() if ln.startswith("[mgp]")}):  # (MGP_TRACE=1)
                print("   ", ln)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[v].append(d["median"])
            sums[v] = (d["csum"], d["vsum"])
    for v in names:
        if res[v]:
            print(f"{v:10s} median-of-medians {sorted(res[v])[len(res[v]) // 2]:8.4f} ms  all {['%.4f' % t for t in res[v]]}  "
                  f"-> {args.b / sorted(res[v])[len(res[v]) // 2] / 1e3:7.1f} M nbhd/s   sums {sums[v]}")


if __name__ == "__main__":
    main()
