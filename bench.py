#!/usr/bin/env python3
"""Headline benchmark: neighbourhoods/s for posterior mean + variance (BASELINE.json).

A "step" is one pass of the fused hot path over one batch of b neighbourhoods with all
inputs resident in HBM: (features, targets, batch_idx, nn_idx) -> (mean, var).  Workload
at N=1 is BASELINE.json configs[1]: Matern-3/2, 1M synthetic points, d=40, nn_count=30,
fp32 (``--config 2``).  ``--config 3`` times one LOOCV objective evaluation of the same
shape (fused launch + fp64 loss sums + the ONE all-reduce of 4+R scalars),
``--config 4`` the anisotropic fp64 k=50 d=8 LOOCV evaluation, ``--config 5`` the k=64, R=16
RBF prediction.

``--gpus N``: one process per GPU over RCCL (torch.distributed backend "nccl").  Started by
``torch.distributed.run`` the ranks come from the environment; started as a plain
``python bench.py --gpus N`` this process (which never touches a GPU) starts N children of
itself with RANK / LOCAL_RANK / WORLD_SIZE set and relays rank 0's line.  Every rank holds a
replica of the tables and its own b neighbourhoods (weak scaling, no data-path collective:
the path shards embarrassingly); the timed region is bracketed by barrier + synchronize on
both sides and the MAX over ranks is reported.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_TFLOPS = 157.3  # same guide: peak FP32 (vector)
FP64_VECTOR_TFLOPS = 78.6   # half the fp32 vector rate (v_fma_f64 at the v_pk_fma_f32 issue cost, tools/ubench)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r02_wave_pmc_traffic.json")
TRAFFIC45_JSON = os.path.join(ROOT, "profiles", "r02_c45_pmc_traffic.json")

# BASELINE.json configs: shape, model and what one step does
CONFIGS = {
    2: dict(points=1_000_000, batch=1_000_000, k=30, d=40, R=1, dtype="f32", kernel="matern15", metric="l2",
            aniso=False, noise=1e-3, objective=False,
            what="posterior mean + variance"),
    3: dict(points=1_000_000, batch=1_000_000, k=30, d=40, R=1, dtype="f32", kernel="matern15", metric="l2",
            aniso=False, noise=1e-3, objective=True,
            what="one LOOCV objective evaluation (mean, variance, y^T K^-1 y -> sigma^2, lool; one all-reduce)"),
    4: dict(points=10_000_000, batch=2_000_000, k=50, d=8, R=1, dtype="f64", kernel="matern15", metric="l2",
            aniso=True, noise=1e-5, objective=True,
            what="one LOOCV objective evaluation of the anisotropic fp64 model (the unit of the Bayes-opt loop)"),
    5: dict(points=2_000_000, batch=500_000, k=64, d=40, R=16, dtype="f32", kernel="rbf", metric="F2",
            aniso=False, noise=1e-3, objective=False,
            what="posterior mean (16 responses) + variance"),
}


def measured_traffic(b: int, k: int, d: int, dtype: str):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
    collected separately, gfx950 correction applied; see profiles/).  The counters were taken on
    the headline shape; scaled by neighbourhood count, else null."""
    if not (k == 30 and d == 40 and dtype == "f32" and os.path.exists(TRAFFIC_JSON)):
        # configs 4 / 5: their own PMC passes (profiles/r02_c45_pmc_traffic.json)
        if os.path.exists(TRAFFIC45_JSON):
            with open(TRAFFIC45_JSON) as f:
                t = json.load(f)["configs"]
            for c, shape in (("4", (50, 8, "f64")), ("5", (64, 40, "f32"))):
                if shape == (k, d, dtype) and c in t:
                    return t[c]["hbm_bytes_per_launch_corrected"] / t[c]["neighbourhoods_per_launch"] * b
        return None
    with open(TRAFFIC_JSON) as f:
        t = json.load(f)
    return t["hbm_bytes_per_launch_corrected"] / t["neighbourhoods_per_launch"] * b


def algorithmic_bytes(k: int, d: int, R: int, s: int, loocv: bool = False) -> int:
    """(k+1) feature rows + k*R neighbour targets + int64 nn row and
    batch index + mean(R) and var out (+ the batch target in LOOCV mode).  5336 B at k=30,
    d=40, R=1, fp32."""
    return (k + 1) * d * s + k * R * s + 8 * (k + 1) + (R + 1) * s + (R * s if loocv else 0)


def algorithmic_flops(k: int, d: int, R: int, loocv: bool = False) -> float:
    """distances [k(k-1)/2 + k] 3d; kernel ~10 per entry; Cholesky k^3/3;
    triangular solves 2k^2 per right-hand side (1 for mean+var, +R for sigma^2); dots 2k(R+1)."""
    pairs = k * (k - 1) / 2 + k
    return pairs * 3 * d + 10 * pairs + k**3 / 3 + 2 * k * k * (1 + (R if loocv else 0)) + 2 * k * (R + 1)


def synth(n: int, d: int, seed: int, R: int = 1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    w = (rng.standard_normal((d, R)) / np.sqrt(d)).astype(np.float32)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal((n, R), dtype=np.float32)
    return X, (y[:, 0] if R == 1 else y).astype(np.float32)


def random_neighbors(n: int, b: int, k: int, seed: int):
    """Uniform random neighbour rows excluding the point itself (cost-equivalent to kNN rows:
    same bytes, same flops; gathers are fully random over the table)."""
    rng = np.random.default_rng(seed)
    bi = rng.permutation(n)[:b].astype(np.int64) if b < n else np.arange(n, dtype=np.int64)
    ni = rng.integers(0, n - 1, size=(b, k), dtype=np.int64)
    ni += ni >= bi[:, None]
    return bi, ni


def knn_neighbors(Xd, bi, k: int, chunk: int = 2048):
    """Exact brute-force kNN on the GPU (self excluded), muygpys_amd/neighbors.py."""
    from muygpys_amd.neighbors import NN_Wrapper

    return NN_Wrapper(Xd, k, chunk=chunk).get_batch_nns(bi)[0]


def _cpu_worker(args):
    """One host process of the P-process CPU baseline (the reference's `mpirun -n P` layout:
    contiguous row blocks, README.md:99-109)."""
    k, d, lo, hi, seed, fp32, chunk = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import muygps_oracle as orc  # checker/baseline only -- never on the product path

    n = 65536
    X, y = synth(n, d, seed)
    ft = np.float32 if fp32 else np.float64
    X, y = X.astype(ft), y.astype(ft)
    rng = np.random.default_rng(seed + 1)
    bi = rng.integers(0, n, size=hi)
    ni = rng.integers(0, n - 1, size=(hi, k))
    ni += ni >= bi[:, None]
    spec = orc.Spec("matern15", "l2", 5.0, 1e-3)
    t0 = time.perf_counter()
    orc.posterior_mean_var_chunked(spec, X, X, bi[lo:hi], ni[lo:hi], y, chunk=chunk)
    return time.perf_counter() - t0


def cpu_baseline(k: int, d: int, sample: int, seed: int):
    """The oracle (numpy restatement of the reference's numpy backend, same op sequence)
    timed on this box's host cores on a bounded sample of the same workload: fp64 and fp32, one
    process and P = os.cpu_count() processes (BASELINE.md sec. 2 / SURVEY.md sec. 8d).  `value`
    is the fp64 single-process figure (the reference's default configuration)."""
    import multiprocessing as mp
    import platform

    cpu = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    P = os.cpu_count() or 1
    variants = {}
    _cpu_worker((k, d, 0, 256, seed, False, 256))  # warm (imports, page-ins)
    for name, fp32, procs in (("fp64_1proc", False, 1), ("fp32_1proc", True, 1), ("fp64_Pproc", False, P),
                              ("fp32_Pproc", True, P)):
        n_s = sample if procs == 1 else sample * min(procs, 16)
        if procs == 1:
            dt = _cpu_worker((k, d, 0, n_s, seed, fp32, 1024))
        else:
            # every process computes its block concurrently; the slowest one's compute time counts
            # (process start-up and the synthetic-data set-up are not part of the reference's timing either)
            bounds = np.linspace(0, n_s, procs + 1).astype(int)
            # one BLAS / OpenMP thread per worker process (set before the children import numpy)
            for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
                os.environ[var] = "1"
            with mp.get_context("spawn").Pool(procs) as pool:
                # small chunks: P processes x the (chunk, k, k, d) difference tensor must fit the host
                dt = max(pool.map(_cpu_worker, [(k, d, int(bounds[i]), int(bounds[i + 1]), seed, fp32, 128)
                                                for i in range(procs)]))
        variants[name] = {"neighborhoods_per_s": n_s / dt, "processes": procs, "sample": n_s, "seconds": dt}
    v = variants["fp64_1proc"]
    return {
        "value": v["neighborhoods_per_s"], "unit": "neighborhoods/s", "cores": 1, "kind": "port",
        "sample": f"{v['sample']} neighbourhoods (k={k}, d={d}, fp64 numpy oracle = the reference's numpy-backend "
                  f"op sequence, chunks of 1024, {v['seconds']:.1f} s)",
        "cpu_model": cpu, "host_cores": P, "numpy": np.__version__,
        "variants": variants,
    }


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """Parent of a plain `python bench.py --gpus N`: N children, one per GPU.  This process makes no
    HIP call (counting devices does not initialise the GPU); it never re-executes itself."""
    import torch

    have = torch.cuda.device_count()
    if have < n and "--one-device" not in argv:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config")
    ap.add_argument("--points", type=int, default=0, help="training points N (replicated per GPU; 0 = the config's)")
    ap.add_argument("--batch", type=int, default=0, help="neighbourhoods per GPU per step (0 = the config's)")
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--d", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "f32", "f64"])
    ap.add_argument("--kernel", default="")
    ap.add_argument("--objective", action="store_true",
                    help="time one LOOCV objective evaluation (fused launch + loss sums + the all-reduce)")
    ap.add_argument("--knn", action="store_true", help="exact GPU kNN neighbourhoods instead of random rows")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="0 disables the CPU baseline leg")
    ap.add_argument("--path", default="auto", choices=["auto", "generic", "rhs"], help="kernel family to time")
    ap.add_argument("--no-prepared-tables", action="store_true", help="read the plain feature / target tables")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for self-tests)")
    ap.add_argument("--one-device", action="store_true",
                    help="self-test only: every rank uses cuda:0 (exercises the N>1 logic on a 1-GPU box)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch

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

    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec, PackedTable, pack_table, posterior_mean_var

    cfg = dict(CONFIGS[args.config])
    for key, val in (("points", args.points), ("batch", args.batch), ("k", args.k), ("d", args.d),
                     ("dtype", args.dtype), ("kernel", args.kernel)):
        if val:
            cfg[key] = val
    if args.objective:
        cfg["objective"] = True
    td = torch.float32 if cfg["dtype"] == "f32" else torch.float64
    n, k, d, R = cfg["points"], cfg["k"], cfg["d"], cfg["R"]
    b = min(cfg["batch"], n)
    X, y = synth(n, d, 20241008, R)  # same table on every rank (replicated)
    Xd = torch.from_numpy(X).to(dev, td)
    yd = torch.from_numpy(y).to(dev, td)
    del X, y
    if args.knn:
        bi_np = np.random.default_rng(1 + rank).permutation(n)[:b].astype(np.int64)
        bi = torch.from_numpy(bi_np).to(dev)
        ni = knn_neighbors(Xd.float(), bi, k)
    else:
        bi_np, ni_np = random_neighbors(n, b, k, 1 + rank)
        bi, ni = torch.from_numpy(bi_np).to(dev), torch.from_numpy(ni_np).to(dev)
    ell = float(np.sqrt(d / 40.0) * 5.0)
    if cfg["metric"] == "F2":
        ell = 5.0
    if cfg["aniso"]:
        ls = list(np.exp(np.random.default_rng(2).uniform(np.log(0.5), np.log(2.0), size=d)) * ell)
    else:
        ls = ell
    spec = KernelSpec(cfg["kernel"], cfg["metric"], ls, cfg["noise"])
    mean = torch.empty((b, R), device=dev, dtype=td)
    var = torch.empty((b,), device=dev, dtype=td)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    # the prepared tables are built once, outside the timed loop (they are constant across all
    # objective evaluations / prediction batches of a model; DESIGN.md sec. 5 gives the pack time)
    use_packed = (not args.no_prepared_tables) and args.path == "auto" and PackedTable.supported(d, R, k, td)
    t_pack = None
    if use_packed:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pack_table(Xd, yd)
        torch.cuda.synchronize()
        t_pack = time.perf_counter() - t0

    if cfg["objective"]:
        def step():
            return D.sharded_loocv(spec, Xd, yd, bi, ni, loss="lool", presharded=True, packed=use_packed)
    else:
        def step():
            posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var, info=info, path=args.path,
                               packed=use_packed)

    last = None
    for _ in range(args.warmup):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        last = step()
        ev[i + 1].record()  # same stream the kernel is enqueued on (torch's current stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    ranks_seen = 1
    if dist is not None:
        cdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=cdev, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # every rank took part in a collective
        ranks_seen = int(ones.item())
    non_spd = int(info.item())
    if cfg["objective"]:
        assert np.isfinite(last["lool"]) and np.isfinite(last["sigma_sq"]), "non-finite objective"
    else:
        assert torch.isfinite(mean).all() and torch.isfinite(var).all(), "non-finite outputs"

    if rank == 0:
        from muygpys_amd import _lib

        s = 4 if cfg["dtype"] == "f32" else 8
        B = algorithmic_bytes(k, d, R, s, loocv=cfg["objective"])
        F = algorithmic_flops(k, d, R, loocv=cfg["objective"])
        avg_ms = float(np.mean(kern_ms))
        achieved = B * b / (avg_ms * 1e-3) / 1e9
        tflops = F * b / (avg_ms * 1e-3) / 1e12
        vpeak = FP32_VECTOR_TFLOPS if cfg["dtype"] == "f32" else FP64_VECTOR_TFLOPS
        kernel_name = _lib.served_by(d, k, R, td, use_packed, args.path)
        out = {
            "metric": "neighborhoods/sec (posterior mean+var)" if not cfg["objective"]
                      else "neighborhoods/sec (LOOCV objective evaluation)",
            "value": world * b * args.steps / elapsed,
            "unit": "neighborhoods/s",
            "n_gpus": world,
            "rccl_ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": cfg["dtype"],
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config}: {cfg['what']}; {cfg['kernel']} nu-fixed, {n} synthetic "
                            f"points, d={d}, nn_count={k}, responses={R}, {b} neighbourhoods per GPU per step, "
                            f"{'exact kNN' if args.knn else 'random'} neighbour rows, "
                            f"{'Anisotropy' if cfg['aniso'] else 'Isotropy'}/{cfg['metric']}, noise={cfg['noise']}",
                "baseline_config": args.config,
                "points": n, "batch_per_gpu": b, "nn_count": k, "feature_count": d, "response_count": R,
                "kernel_path": args.path, "prepared_tables": bool(use_packed),
                "prepared_table_pack_ms": None if t_pack is None else t_pack * 1e3,
                "non_spd_neighbourhoods": non_spd,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(b, k, d, cfg["dtype"]),
                "kernel": kernel_name,
                "algorithmic_bytes_per_neighbourhood": B, "algorithmic_bytes_per_launch": B * b,
                "kernel_ms": avg_ms,
                "valu": {"achieved": tflops, "peak": vpeak, "unit": "TFLOP/s", "frac": tflops / vpeak,
                         "algorithmic_flops_per_neighbourhood": F},
            },
        }
        if cfg["objective"]:
            out["config"]["lool"] = last["lool"]
            out["config"]["sigma_sq"] = last["sigma_sq"]
            out["roofline"]["note"] = ("kernel_ms is the whole step (fused launch + fp64 loss reductions + host finish "
                                       "+ all-reduce), so achieved is a lower bound for the fused kernel alone")
        if args.cpu_sample > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(k, d, args.cpu_sample, 20241008)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
