#!/usr/bin/env python3
"""Timing of the exact GPU k-NN index producer (GPU box only).

    python tools/knnbench.py [--n 1000000] [--d 40] [--k 30] [--queries 100000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muygpys_amd.neighbors import NN_Wrapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--scan", default="auto", choices=["auto", "bf16x3", "f32", "dense"])
    args = ap.parse_args()
    torch.manual_seed(0)
    X = torch.randn(args.n, args.d, device="cuda")
    nn = NN_Wrapper(X, args.k, chunk=args.chunk, use_scan=args.scan != "dense",
                    scan_kind=args.scan if args.scan != "dense" else "f32")
    bi = torch.arange(args.queries, device="cuda")
    nn.get_batch_nns(bi[:8192])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, dist = nn.get_batch_nns(bi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = args.queries * args.n
    print(f"[{args.scan}] kNN {args.queries} queries x {args.n} points, d={args.d}, k={args.k}: {dt * 1e3:.1f} ms "
          f"-> {args.queries / dt / 1e3:.1f} k queries/s, {pairs / dt / 1e12:.2f} T pair-distances/s, "
          f"{2 * pairs * args.d / dt / 1e12:.1f} TFLOP/s")
    # (for A/B runs of kernel variants: the same lists whatever the tiling)
    print(f"checksum: indices {int(idx.sort(dim=1).values.to(torch.int64).sum().item())}, distances {float(dist.double().sum().item()):.9e}")


if __name__ == "__main__":
    main()
