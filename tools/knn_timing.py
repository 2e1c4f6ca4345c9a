#!/usr/bin/env python3
"""Where a wave of the split-bf16 k-NN scan spends its cycles (GPU box; a library built with -DMGP_KNN_TIMING=1:
tools/mkvariant.sh NAME mgp_knn.hip -DMGP_KNN_TIMING=1, MUYGPYS_HIP_LIB=variants/lib_NAME.so).

    python tools/knn_timing.py [--n 10000000 --d 8 --k 50 --queries 400000]
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from muygpys_amd import _lib
from muygpys_amd.neighbors import NN_Wrapper

PHASES = ["tile wait + barrier", "tile issue", "matrix blocks + survivor queues", "drain"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--queries", type=int, default=400_000)
    args = ap.parse_args()
    lib = _lib.load()
    fn = lib.mgp_debug_knn_timing
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    torch.manual_seed(0)
    X = torch.randn(args.n, args.d, device="cuda")
    nn = NN_Wrapper(X, args.k)
    bi = torch.arange(args.queries, device="cuda")
    nn.get_batch_nns(bi[:8192])
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    fn(out, 1)
    t0 = time.perf_counter()
    nn.get_batch_nns(bi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fn(out, 0)
    total = float(sum(out[:4]))
    print(f"{args.queries} x {args.n}, d = {args.d}, k = {args.k}: {dt * 1e3:.1f} ms")
    for name, v in zip(PHASES, out[:4]):
        print(f"  {name:34s} {100.0 * v / total:5.1f} %   {v / 1e9:9.2f} G cycles summed over waves")


if __name__ == "__main__":
    main()
