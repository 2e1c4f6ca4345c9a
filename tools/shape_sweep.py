#!/usr/bin/env python3
"""Shape sweep of the fused posterior (GPU box): k x d grid, one precision, prepared tables.

    python tools/shape_sweep.py [--dtype f32] [--ks 10,20,25,30,40,50] [--ds 8,16,32,40,64] [--b 500000] [--md out.md]

Per shape: M neighbourhoods/s, algorithmic TFLOP/s (bench.algorithmic_flops), the per-flop rate relative to
the headline shape (k = 30, d = 40) and the kernel instantiation that served the call.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import algorithmic_flops, random_neighbors, synth
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, clear_caches, posterior_mean_var


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--ks", default="10,20,25,30,40,50")
    ap.add_argument("--ds", default="8,16,32,40,64")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--b", type=int, default=500_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--md", default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    td = torch.float32 if args.dtype == "f32" else torch.float64
    ks = [int(v) for v in args.ks.split(",")]
    ds = [int(v) for v in args.ds.split(",")]
    rows = []
    for d in ds:
        X, y = synth(args.n, d, 20241008, 1)
        Xd, yd = torch.from_numpy(X).to(dev, td), torch.from_numpy(y).to(dev, td)
        for k in ks:
            bi, ni = random_neighbors(args.n, args.b, k, 1)
            bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
            spec = KernelSpec("matern15", "l2", float(np.sqrt(d / 40.0) * 5.0), 1e-3)
            mean = torch.empty((args.b, 1), device=dev, dtype=td)
            var = torch.empty((args.b,), device=dev, dtype=td)
            ts = []
            for r in range(args.rounds + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, packed=True)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    ts.append(e0.elapsed_time(e1))
            assert torch.isfinite(mean).all() and torch.isfinite(var).all()
            ms = float(np.median(ts))
            F = algorithmic_flops(k, d, 1)
            rows.append(dict(k=k, d=d, ms=ms, mnbhd=args.b / ms / 1e3, tflops=F * args.b / ms / 1e9,
                             kernel=_lib.last_kernel().replace("mgp::", "")))
            print(rows[-1], flush=True)
        del Xd, yd
        clear_caches()
        torch.cuda.empty_cache()
    head = next((r for r in rows if r["k"] == 30 and r["d"] == 40), None)
    lines = [f"| k | d | ms per {args.b} | M nbhd/s | TFLOP/s (algorithmic) | per-flop rate vs k=30,d=40 | kernel |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        rel = "" if head is None else f"{r['tflops'] / head['tflops']:.2f}"
        lines.append(f"| {r['k']} | {r['d']} | {r['ms']:.3f} | {r['mnbhd']:.1f} | {r['tflops']:.2f} | {rel} | `{r['kernel']}` |")
    text = "\n".join(lines)
    print(text)
    if args.md:
        with open(args.md, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
