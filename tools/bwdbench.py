#!/usr/bin/env python3
"""Timing of the backward (vector-Jacobian) kernel next to the forward launch (GPU box only).

    python tools/bwdbench.py [--b 1000000] [--k 30] [--d 40] [--dtype f32] [--outs x,ls,noise,y]
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
    ap.add_argument("--R", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--aniso", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--outs", default="x,ls,noise,y;x;ls,noise")
    ap.add_argument("--stages", default="0", help="comma list of mgp_debug_set_bwd_stage values (0 = whole kernel)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    td = torch.float32 if args.dtype == "f32" else torch.float64
    X, y = synth(args.n, args.d, 20241008)
    Xd, yd = torch.from_numpy(X).to(dev, td), torch.from_numpy(y).to(dev, td)[:, None].repeat(1, args.R).contiguous()
    bi, ni = random_neighbors(args.n, args.b, args.k, 1)
    bi, ni = torch.from_numpy(bi).to(dev), torch.from_numpy(ni).to(dev)
    lsv = torch.full((args.d if args.aniso else 1,), 5.0, device=dev, dtype=td)
    gm = torch.randn(args.b, args.R, device=dev, dtype=td)
    gv = torch.randn(args.b, device=dev, dtype=td)
    gx, gy = torch.zeros_like(Xd), torch.zeros_like(yd)
    gl = torch.empty(args.b, lsv.numel(), device=dev, dtype=td)
    gn = torch.empty(args.b, args.k, device=dev, dtype=td)
    info = torch.zeros(1, device=dev, dtype=torch.int32)
    P = _lib.ptr
    fn = _lib.fn("posterior_backward", td)
    spec = KernelSpec("matern15", "l2", [5.0] * args.d if args.aniso else 5.0, 1e-3)

    def fwd():
        posterior_mean_var(spec, Xd, Xd, bi, ni, yd)

    def bwd(outs):
        o = set(outs.split(","))
        rc = fn(P(Xd), P(Xd), args.d, P(bi), P(ni), args.b, args.k, P(yd), args.R, 0, 1e-3, None, 2, 0, P(lsv),
                lsv.numel(), P(gm), P(gv), P(gx) if "x" in o else None, P(gx) if "x" in o else None,
                P(gy) if "y" in o else None, P(gl) if "ls" in o else None, P(gn) if "noise" in o else None,
                P(info), _lib.stream_ptr())
        assert rc == 0, rc

    lib = _lib.load()

    def staged(o, st):
        if st or hasattr(lib, "mgp_debug_set_bwd_stage"):
            lib.mgp_debug_set_bwd_stage(st)  # stages need a build with -DMGP_DEBUG_HOOKS
        bwd(o)
        if hasattr(lib, "mgp_debug_set_bwd_stage"):
            lib.mgp_debug_set_bwd_stage(0)

    variants = [("forward", fwd)] + [
        (f"backward[{o}] stage={st}", (lambda o=o, st=st: staged(o, st)))
        for o in args.outs.split(";") for st in map(int, args.stages.split(","))
    ]
    times = {name: [] for name, _ in variants}
    for r in range(args.rounds + 1):
        for name, f in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1))
    for name, _ in variants:
        t = np.median(times[name])
        print(f"{name:36s} median {t:9.3f} ms -> {args.b / t / 1e3:8.1f} M nbhd/s")


if __name__ == "__main__":
    main()
