#!/usr/bin/env python3
"""Interleaved in-process A/B timing of the fused kernel variants (GPU box only).

    python tools/kbench.py [--b 1000000] [--rounds 5] masks=15,1,3,7 generic=0,1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import random_neighbors, synth
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--b", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--masks", default="15")
    ap.add_argument("--generic", default="0")
    ap.add_argument("--aniso", type=int, default=0)
    ap.add_argument("--R", type=int, default=1)
    ap.add_argument("--kernel", default="matern15")
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--grids", default="0")
    ap.add_argument("--ldspad", type=int, default=0)
    ap.add_argument("--wave2", default="0")
    ap.add_argument("--rtpipe", default="1", help="comma list: 1 = pipelined direct-to-LDS gather for run-time shapes")
    ap.add_argument("--prefer-rhs", type=int, default=0)
    ap.add_argument("--hot-rows", type=int, default=0, help="restrict neighbour rows to the first N (cache-resident gather)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    td = torch.float32 if args.dtype == "f32" else torch.float64
    X, y = synth(args.n, args.d, 20241008)
    Xd, yd = torch.from_numpy(X).to(dev, td), torch.from_numpy(y).to(dev, td)
    if args.R > 1:
        yd = yd[:, None].repeat(1, args.R).contiguous() * torch.linspace(0.5, 1.5, args.R, device=dev, dtype=td)
    bi, ni = random_neighbors(args.n, args.b, args.k, 1)
    bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
    if args.hot_rows:
        ni = ni % args.hot_rows
        bi = bi % args.hot_rows + args.hot_rows
    ls = [5.0] * args.d if args.aniso else 5.0
    spec = KernelSpec(args.kernel, args.metric, ls, 1e-3)
    mean = torch.empty((args.b, args.R), device=dev, dtype=td)
    var = torch.empty((args.b,), device=dev, dtype=td)
    lib = _lib.load()
    lib.mgp_debug_set_lds_pad(args.ldspad)
    lib.mgp_debug_prefer_rhs(args.prefer_rhs)
    variants = [(int(m), int(g), int(pc), int(w2), int(rp)) for g in args.generic.split(",")
                for m in args.masks.split(",") for pc in args.grids.split(",") for w2 in args.wave2.split(",")
                for rp in args.rtpipe.split(",")]
    times = {v: [] for v in variants}
    for r in range(args.rounds + 1):
        for v in variants:
            lib.mgp_debug_set_phase_mask(v[0])
            lib.mgp_debug_force_generic(v[1])
            lib.mgp_debug_set_grid_per_cu(v[2])
            lib.mgp_debug_enable_wave2(v[3])
            lib.mgp_debug_runtime_pipe(v[4])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1))
    for v in variants:
        t = np.array(times[v])
        print(f"mask={v[0]:2d} generic={v[1]} grid/cu={v[2]:2d} wave2={v[3]} rtpipe={v[4]} median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  "
              f"-> {args.b / np.median(t) / 1e3:8.1f} M nbhd/s")


if __name__ == "__main__":
    main()
