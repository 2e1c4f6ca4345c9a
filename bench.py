#!/usr/bin/env python3
"""Headline benchmark: neighbourhoods/s for posterior mean + variance (BASELINE.json).

A "step" is one pass of the fused hot path over one batch of b neighbourhoods with all
inputs resident in HBM: (features, targets, batch_idx, nn_idx) -> (mean, var).  Workload
at N=1 is BASELINE.json configs[1]: Matern-3/2, 1M synthetic points, d=40, nn_count=30,
fp32.  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank
holds a replica of the feature/target tables and its own b neighbourhoods (weak scaling,
no data-path collective -- the path shards embarrassingly); the only
collective is the barrier + MAX-reduce of the elapsed time.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r01_wave_pmc_traffic.json")


def measured_traffic(b: int, k: int, d: int, dtype: str):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
    collected separately, gfx950 correction applied; see profiles/r01_wave_pmc_traffic.json).
    The counters were taken on the headline shape; scaled by neighbourhood count, else null."""
    if not (k == 30 and d == 40 and dtype == "f32" and os.path.exists(TRAFFIC_JSON)):
        return None
    with open(TRAFFIC_JSON) as f:
        t = json.load(f)
    return t["hbm_bytes_per_launch_corrected"] / t["neighbourhoods_per_launch"] * b


def algorithmic_bytes(k: int, d: int, R: int, s: int) -> int:
    """(k+1) feature rows + k*R neighbour targets + int64 nn row and
    batch index + mean(R) and var out.  5336 B at k=30, d=40, R=1, fp32."""
    return (k + 1) * d * s + k * R * s + 8 * (k + 1) + (R + 1) * s


def synth(n: int, d: int, seed: int):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    w = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n, dtype=np.float32)
    return X, y.astype(np.float32)


def random_neighbors(n: int, b: int, k: int, seed: int):
    """Uniform random neighbour rows excluding the point itself (cost-equivalent to kNN rows:
    same bytes, same flops; gathers are fully random over the table)."""
    rng = np.random.default_rng(seed)
    bi = rng.permutation(n)[:b].astype(np.int64) if b < n else np.arange(n, dtype=np.int64)
    ni = rng.integers(0, n - 1, size=(b, k), dtype=np.int64)
    ni += ni >= bi[:, None]
    return bi, ni


def knn_neighbors(Xd: torch.Tensor, bi: torch.Tensor, k: int, chunk: int = 2048):
    """Exact brute-force kNN on the GPU (self excluded), muygpys_amd/neighbors.py."""
    from muygpys_amd.neighbors import NN_Wrapper

    return NN_Wrapper(Xd, k, chunk=chunk).get_batch_nns(bi)[0]


def cpu_baseline(k: int, d: int, sample: int, seed: int):
    """The oracle (numpy restatement of the reference's numpy backend, same op sequence)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import muygps_oracle as orc  # checker/baseline only -- never on the product path

    n = max(20000, sample)
    X, y = synth(n, d, seed)
    X, y = X.astype(np.float64), y.astype(np.float64)
    bi, ni = random_neighbors(n, sample, k, seed + 1)
    spec = orc.Spec("matern15", "l2", 5.0, 1e-3)
    orc.posterior_mean_var_chunked(spec, X, X, bi[:256], ni[:256], y, chunk=256)  # warm
    t0 = time.perf_counter()
    orc.posterior_mean_var_chunked(spec, X, X, bi, ni, y, chunk=1024)
    dt = time.perf_counter() - t0
    return {
        "value": sample / dt, "unit": "neighborhoods/s", "cores": 1, "kind": "port",
        "sample": f"{sample} neighbourhoods (k={k}, d={d}, fp64 numpy oracle, chunks of 1024, {dt:.1f} s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000, help="training points N (replicated per GPU)")
    ap.add_argument("--batch", type=int, default=0, help="neighbourhoods per GPU per step (0 = N)")
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--d", type=int, default=40)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--kernel", default="matern15")
    ap.add_argument("--knn", action="store_true", help="exact GPU kNN neighbourhoods instead of random rows")
    ap.add_argument("--cpu-sample", type=int, default=131072, help="0 disables the CPU baseline leg")
    ap.add_argument("--force-generic", action="store_true", help="time the generic LDS kernel")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for self-tests)")
    ap.add_argument("--one-device", action="store_true",
                    help="self-test only: every rank uses cuda:0 (exercises the N>1 logic on a 1-GPU box with gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the hip path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    td = torch.float32 if args.dtype == "f32" else torch.float64
    n, k, d = args.points, args.k, args.d
    b = args.batch or n
    X, y = synth(n, d, 20241008)  # same table on every rank (replicated)
    Xd = torch.from_numpy(X).to(dev, td)
    yd = torch.from_numpy(y).to(dev, td)
    if args.knn:
        bi_np = np.random.default_rng(1 + rank).permutation(n)[:b].astype(np.int64)
        bi = torch.from_numpy(bi_np).to(dev)
        ni = knn_neighbors(Xd.float(), bi, k)
    else:
        bi_np, ni_np = random_neighbors(n, b, k, 1 + rank)
        bi, ni = torch.from_numpy(bi_np).to(dev), torch.from_numpy(ni_np).to(dev)
    spec = KernelSpec(args.kernel, "l2", 5.0, 1e-3)
    mean = torch.empty((b, 1), device=dev, dtype=td)
    var = torch.empty((b,), device=dev, dtype=td)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.load().mgp_debug_force_generic(1 if args.force_generic else 0)

    def step():
        posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, info=info)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()  # same stream the kernel is enqueued on (torch's current stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    non_spd = int(info.item())
    assert torch.isfinite(mean).all() and torch.isfinite(var).all(), "non-finite outputs"

    if rank == 0:
        s = 4 if args.dtype == "f32" else 8
        B = algorithmic_bytes(k, d, 1, s)
        avg_ms = float(np.mean(kern_ms))
        achieved = B * b / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "neighborhoods/sec (posterior mean+var)",
            "value": world * b * args.steps / elapsed,
            "unit": "neighborhoods/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.kernel} nu-fixed, {n} synthetic points, d={d}, nn_count={k}, "
                            f"{b} neighbourhoods per GPU per step, {'exact kNN' if args.knn else 'random'} neighbour rows, "
                            "Isotropy/l2 length_scale=5.0, noise=1e-3",
                "points": n, "batch_per_gpu": b, "nn_count": k, "feature_count": d, "response_count": 1,
                "kernel_path": "generic-lds" if args.force_generic else "auto",
                "non_spd_neighbourhoods": non_spd,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(b, k, d, args.dtype),
                "kernel": "mgp::fused_wave_kernel" if not args.force_generic else "mgp::fused_generic_kernel",
                "algorithmic_bytes_per_neighbourhood": B, "algorithmic_bytes_per_launch": B * b,
                "kernel_ms": avg_ms,
            },
        }
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(k, d, args.cpu_sample, 20241008)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
