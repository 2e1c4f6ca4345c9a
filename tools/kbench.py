#!/usr/bin/env python3
"""Interleaved in-process A/B timing of the fused kernel families (GPU box only).

    python tools/kbench.py --k 100 --d 40 --paths auto,generic [--b 200000] [--rounds 5]

``--paths``: auto (dispatcher), generic (LDS workgroup kernel), rhs (responses as columns);
``--packed``: 0/1 prepared tables (auto path only).  The phase-mask / grid ablations of round 1 need a
debug build of the library (MGP_EXTRA_HIPCC_FLAGS=-DMGP_DEBUG_HOOKS) and ``--masks``.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import algorithmic_flops, random_neighbors, synth
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--b", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--R", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--paths", default="auto")
    ap.add_argument("--packed", default="1")
    ap.add_argument("--aniso", type=int, default=0)
    ap.add_argument("--kernel", default="matern15")
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--masks", default="", help="debug builds only: comma list of phase masks")
    ap.add_argument("--hot-rows", type=int, default=0, help="restrict neighbour rows to the first N (cache-resident gather)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    td = torch.float32 if args.dtype == "f32" else torch.float64
    X, y = synth(args.n, args.d, 20241008, args.R)
    Xd, yd = torch.from_numpy(X).to(dev, td), torch.from_numpy(y).to(dev, td)
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
    masks = [int(m) for m in args.masks.split(",")] if args.masks else [None]
    if args.masks and not hasattr(lib, "mgp_debug_set_phase_mask"):
        raise SystemExit("--masks needs a library built with -DMGP_DEBUG_HOOKS")
    variants = [(p, int(pk), m) for p in args.paths.split(",") for pk in (args.packed.split(",") if p == "auto" else ["0"])
                for m in masks]
    times = {v: [] for v in variants}
    for r in range(args.rounds + 1):
        for v in variants:
            if v[2] is not None:
                lib.mgp_debug_set_phase_mask(v[2])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, path=v[0], packed=bool(v[1]))
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1))
    flops = algorithmic_flops(args.k, args.d, args.R)
    for v in variants:
        t = np.array(times[v])
        name = _lib.served_by(args.d, args.k, args.R, td, bool(v[1]), v[0])
        print(f"path={v[0]:8s} packed={v[1]} mask={v[2]} median {np.median(t):9.3f} ms  min {t.min():9.3f} ms -> "
              f"{args.b / np.median(t) / 1e3:8.1f} M nbhd/s  {flops * args.b / np.median(t) / 1e9:7.2f} TFLOP/s  [{name}]")


if __name__ == "__main__":
    main()
