#!/usr/bin/env python3
"""Bandwidth of the materialising per-function kernels (mgp_tensor_ops.hip) on the GPU box:
algorithmic bytes (inputs read once + outputs written once) / HIP-event time, next to the 8 TB/s
HBM peak.  These kernels exist for API parity (a caller who asks for the intermediate tensors gets
them); the hot path never runs them.

    python tools/opbench.py [--b 200000] [--k 30] [--d 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from muygpys_amd import _lib
from muygpys_amd._src.gp.kernels import hip as K
from muygpys_amd._src.gp.noise import hip as N
from muygpys_amd._src.gp.tensors import hip as T
from muygpys_amd.fused import PackedTable


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--b", type=int, default=200_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--d", type=int, default=40)
    args = ap.parse_args()
    n, b, k, d, s = args.n, args.b, args.k, args.d, 4
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(n, d, device="cuda", generator=g)
    y = torch.randn(n, device="cuda", generator=g)
    bi = torch.randint(0, n, (b,), device="cuda", generator=g)
    ni = torch.randint(0, n, (b, k), device="cuda", generator=g)
    rows = []
    cw = T._crosswise_tensor_now(X, X, bi, ni)
    rows.append(("mgp_crosswise_diffs", timed(lambda: T._crosswise_tensor_now(X, X, bi, ni)), b * (k + 1) * d * s + b * k * d * s + 8 * b * (k + 1)))
    bp = b // 8
    pw = T._pairwise_tensor_now(X, ni[:bp])
    rows.append(("mgp_pairwise_diffs", timed(lambda: T._pairwise_tensor_now(X, ni[:bp])), bp * k * d * s + bp * k * k * d * s + 8 * bp * k))
    rows.append(("mgp_pairwise_dists (fused gather + metric)", timed(lambda: T._pairwise_distances(X, ni, "l2")), b * k * d * s + b * k * k * s + 8 * b * k))
    rows.append(("mgp_crosswise_dists", timed(lambda: T._crosswise_distances(X, X, bi, ni, "l2")), b * (k + 1) * d * s + b * k * s + 8 * b * (k + 1)))
    rows.append(("mgp_reduce_diffs (l2)", timed(lambda: T._reduce(pw, 0)), pw.numel() * s + pw.numel() // d * s))
    dist = T._pairwise_distances(X, ni, "l2")
    rows.append(("mgp_kernel_apply (matern15)", timed(lambda: K._apply(dist, "matern15", 0.2)), 2 * dist.numel() * s))
    rows.append(("mgp_matern_gen (nu = 0.42, fp64 Bessel K)", timed(lambda: K._matern_gen_fn(dist * 0.2, 0.42)), 2 * dist.numel() * s))
    Kin = K._apply(dist, "matern15", 0.2)
    rows.append(("mgp_perturb", timed(lambda: N._homoscedastic_perturb(Kin, 1e-3)), 2 * Kin.numel() * s))
    from muygpys_amd._src.gp.muygps import hip as M
    Kp = N._homoscedastic_perturb(Kin, 1e-3)
    Kc = K._apply(T._crosswise_distances(X, X, bi, ni, "l2"), "matern15", 0.2)
    ynn = y[ni]
    rows.append(("mgp_solve (posterior mean on materialised Kin)", timed(lambda: M._muygps_posterior_mean(Kp, Kc, ynn)),
                 b * (k * k + 2 * k + 1) * s))
    rows.append(("mgp_solve (diagonal variance)", timed(lambda: M._muygps_diagonal_variance(Kp, Kc, 1.0)), b * (k * k + k + 1) * s))
    v = torch.rand(n, device="cuda") + 0.1
    rows.append(("mgp_loss_sums (deterministic)", timed(lambda: _lib.loss_sums(y, y * 0.9, v, None, 1.5, 3.0)), 3 * n * s))
    y2 = torch.randn(n, 4, device="cuda", generator=g)
    rows.append(("mgp_column_sums (R = 4, deterministic)", timed(lambda: _lib.column_sums(y2)), y2.numel() * s))
    rows.append(("mgp_table_pack", timed(lambda: PackedTable(X, y)), n * (d + 1) * s + n * 192))
    print(f"| kernel | ms | algorithmic GB | GB/s | % of 8 TB/s |\n|---|---|---|---|---|")
    for name, ms, nbytes in rows:
        print(f"| {name} | {ms:.3f} | {nbytes / 1e9:.3f} | {nbytes / ms / 1e6:.0f} | {nbytes / ms / 1e6 / 80:.1f} |")


if __name__ == "__main__":
    main()
