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
both sides and the MAX over ranks is reported.  ``--scaling strong`` shards ONE global batch of the
config's size over the ranks instead (reference chunk rule, _src/mpi_utils.py:36-41: BASELINE config 3's
"same 1M batch sharded across 8 GPUs"); the line says which of the two was run.

After every timed loop (headline and each secondary) 256 of the outputs the loop left in HBM are compared with
the fp64 LDS workgroup kernel (an independent kernel family, plain tables) and the largest error goes into the
line as ``check``: a number from a kernel whose results are wrong is not a number.

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
# committed PMC passes, newest first (each names the shape(s) it was taken on)
TRAFFIC_FILES = [(os.path.join(ROOT, "profiles", f), f) for f in (
    "r06_wave_pmc_traffic.json", "r06_c45_pmc_traffic.json", "r05_wave_pmc_traffic.json", "r05_c45_pmc_traffic.json", "r04_wave_pmc_traffic.json", "r04_c45_pmc_traffic.json", "r03_wave_pmc_traffic.json", "r03_c45_pmc_traffic.json",
    "r02_wave_pmc_traffic.json", "r02_c45_pmc_traffic.json")]

# BASELINE.json configs: shape, model and what one step does
CONFIGS = {
    2: dict(points=1_000_000, batch=1_000_000, k=30, d=40, R=1, dtype="f32", kernel="matern15", metric="l2",
            aniso=False, noise=1e-3, objective=False,
            what="posterior mean + variance"),
    3: dict(points=1_000_000, batch=1_000_000, k=30, d=40, R=1, dtype="f32", kernel="matern15", metric="l2",
            aniso=False, noise=1e-3, objective=True,
            what="one LOOCV objective evaluation (mean, variance, y^T K^-1 y -> sigma^2, lool; one all-reduce)"),
    4: dict(points=10_000_000, batch=10_000_000, k=50, d=8, R=1, dtype="f64", kernel="matern15", metric="l2",
            aniso=True, noise=1e-5, objective=True,
            what="one LOOCV objective evaluation of the anisotropic fp64 model (the unit of the Bayes-opt loop)"),
    5: dict(points=2_000_000, batch=500_000, k=64, d=40, R=16, dtype="f32", kernel="rbf", metric="F2",
            aniso=False, noise=1e-3, objective=False,
            what="posterior mean (16 responses) + variance"),
}


def measured_traffic(b: int, k: int, d: int, dtype: str):
    """(HBM-side bytes per launch, where they come from): the committed rocprofv3 PMC passes (FETCH_SIZE /
    WRITE_SIZE collected separately, gfx950 correction applied; see profiles/), scaled by neighbourhood
    count -- NOT measured in this run; (None, None) for shapes without a pass."""
    for path, key in TRAFFIC_FILES:
        if not os.path.exists(path):
            continue
        with open(path) as f:
            t = json.load(f)
        legacy = {"2": (30, 40, "f32"), "4": (50, 8, "f64"), "5": (64, 40, "f32")}  # files without a "shape" entry
        for cid, c in (t.get("configs") or {"2": t}).items():
            shape = c.get("shape")
            shape = (shape["k"], shape["d"], shape["dtype"]) if shape else legacy.get(cid)
            if shape == (k, d, dtype):
                per = c["hbm_bytes_per_launch_corrected"] / c["neighbourhoods_per_launch"]
                return per * b, os.path.relpath(path, ROOT) + " (rocprofv3 --pmc, committed; not measured in this run)"
    return None, None


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
    """One host process of the CPU baseline: rows [lo, hi) of a batch of `hi` neighbourhoods on an n-row table
    (P processes: the reference's `mpirun -n P` layout, contiguous row blocks, README.md:99-109).  Returns
    (seconds of the evaluation itself, seconds since this function was entered)."""
    k, d, lo, hi, seed, fp32, chunk, n = args
    t_in = time.perf_counter()
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import muygps_oracle as orc  # checker/baseline only -- never on the product path

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
    t1 = time.perf_counter()
    return t1 - t0, t1 - t_in


CPU_BATCH, CPU_CHUNK, CPU_RUNS = 65536, 4096, 3


def cpu_baseline(k: int, d: int, sample: int, seed: int, points: int = 1_000_000):
    """BASELINE.md sec. 2, as stated there: the oracle (numpy restatement of the reference's numpy backend, same op
    sequence) on this box's host cores -- b = 65 536 neighbourhoods of the config-2 shape on the 1 M-row table in
    chunks of 4 096, fp64 and fp32, one process: MEDIAN OF 3 runs each; and P = os.cpu_count() worker processes over
    contiguous row blocks (one run each, the compute time of the slowest worker AND the wall clock of the whole pool
    with process start-up and table set-up).  `value` is the fp64 single-process median (the reference's default
    configuration).  `sample` < 65 536 (--cpu-sample) shrinks the batch for self-tests and says so."""
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
    b1 = min(sample, CPU_BATCH)
    chunk1 = min(CPU_CHUNK, b1)
    variants = {}
    _cpu_worker((k, d, 0, 256, seed, False, 256, 4096))  # warm (imports, page-ins)
    for name, fp32 in (("fp64_1proc", False), ("fp32_1proc", True)):
        runs = sorted(_cpu_worker((k, d, 0, b1, seed, fp32, chunk1, points))[0] for _ in range(CPU_RUNS))
        dt = runs[len(runs) // 2]
        variants[name] = {"neighborhoods_per_s": b1 / dt, "processes": 1, "sample": b1, "chunk": chunk1, "table_rows": points,
                          "seconds_median": dt, "seconds_runs": runs}
    if P > 1:
        # every worker: its block of 65 536 x min(P, 8) neighbourhoods; the (chunk, k, k, d) difference tensors of all
        # workers together are held to ~64 GB of host memory (P = 256: chunks of 512 instead of 4 096), and each worker
        # builds its own copy of a 131 072-row table (1 M rows x 256 processes would be 80 GB of tables)
        n_s = b1 * min(P, 8)
        per = -(-n_s // P)
        cap = int(64e9 // (P * k * k * d * 8))
        chunkP = int(max(128, min(CPU_CHUNK, per, 1 << max(cap, 1).bit_length() - 1)))
        bounds = np.linspace(0, n_s, P + 1).astype(int)
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # (before the children import numpy)
            os.environ[var] = "1"
        for name, fp32 in (("fp64_Pproc", False), ("fp32_Pproc", True)):
            t0 = time.perf_counter()
            with mp.get_context("spawn").Pool(P) as pool:
                res = pool.map(_cpu_worker, [(k, d, int(bounds[i]), int(bounds[i + 1]), seed, fp32, chunkP, 131072)
                                             for i in range(P)])
            wall = time.perf_counter() - t0
            dt = max(r[0] for r in res)
            variants[name] = {"neighborhoods_per_s": n_s / dt, "neighborhoods_per_s_with_startup": n_s / wall,
                              "processes": P, "sample": n_s, "chunk": chunkP, "table_rows": 131072,
                              "seconds": dt, "seconds_with_startup": wall}
    v = variants["fp64_1proc"]
    return {
        "value": v["neighborhoods_per_s"], "unit": "neighborhoods/s", "cores": 1, "kind": "port",
        "sample": f"{v['sample']} neighbourhoods (k={k}, d={d}) on a {points}-row table in chunks of {v['chunk']}, fp64 numpy "
                  f"oracle = the reference's numpy-backend op sequence, median of {CPU_RUNS} runs ({v['seconds_median']:.1f} s each; "
                  f"BASELINE.md sec. 2)",
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
    return supervise_ranks(procs, float(os.environ.get("BENCH_RANKS_TIMEOUT_S", "1800")))


def supervise_ranks(procs, timeout_s: float, poll_s: float = 0.05, grace_s: float = 5.0) -> int:
    """Wait for the ranks TOGETHER: the first one to exit non-zero takes its siblings down at once (a rank that dies
    before its first collective would otherwise leave the others blocked in RCCL until the watchdog, ten minutes),
    and so does the overall timeout.  Returns 0 only if every rank returned 0; the first failing rank's code
    otherwise (124 for the timeout, as coreutils' timeout does).  Children are terminated, then killed after a grace
    period -- by handle, never by pattern."""
    deadline = time.monotonic() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0] if bad[0] > 0 else 128 - bad[0]  # (a signal's negative code -> the shell's 128 + signal)
            print(f"bench.py: a rank exited with {bad[0]}; stopping the other ranks", file=sys.stderr)
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > deadline:
            rc = 124
            print(f"bench.py: ranks still running after {timeout_s:.0f} s; stopping them", file=sys.stderr)
            break
        time.sleep(poll_s)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def build_workload(cfg, dev, rank: int, knn: bool, world: int = 1, strong: bool = False):
    """Synthetic tables + neighbourhood indices of one config, resident in HBM.  Weak scaling: every rank draws
    its own b neighbourhoods; strong scaling: every rank draws the SAME global batch of b neighbourhoods and keeps
    its contiguous block (the reference's chunk rule, _src/mpi_utils.py:36-41)."""
    import torch

    from muygpys_amd.distributed import shard_bounds

    td = torch.float32 if cfg["dtype"] == "f32" else torch.float64
    n, k, d, R = cfg["points"], cfg["k"], cfg["d"], cfg["R"]
    b = min(cfg["batch"], n)
    X, y = synth(n, d, 20241008, R)  # same table on every rank (replicated)
    Xd = torch.from_numpy(X).to(dev, td)
    yd = torch.from_numpy(y).to(dev, td)
    del X, y
    lo, hi = shard_bounds(b, rank, world) if strong else (0, b)
    seed = 1 if strong else 1 + rank
    if knn:
        bi_np = np.random.default_rng(seed).permutation(n)[:b].astype(np.int64)[lo:hi]
        bi = torch.from_numpy(bi_np).to(dev)
        ni = knn_neighbors(Xd.float(), bi, k)
    elif b * k >= 100_000_000 and dev.type == "cuda":
        # (config 4 at b = N = 10 M: 4 GB of int64 neighbour rows -- drawn on the device, same distribution as
        # random_neighbors: uniform rows of the table excluding the point itself)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        bi = (torch.randperm(n, generator=g, device=dev)[:b] if b < n else torch.arange(n, device=dev))[lo:hi].contiguous()
        ni = torch.randint(0, n - 1, (hi - lo, k), generator=g, device=dev)
        ni += ni >= bi[:, None]
    else:
        bi_np, ni_np = random_neighbors(n, b, k, seed)
        bi, ni = torch.from_numpy(bi_np[lo:hi].copy()).to(dev), torch.from_numpy(ni_np[lo:hi].copy()).to(dev)
    b = hi - lo
    ell = float(np.sqrt(d / 40.0) * 5.0)
    if cfg["metric"] == "F2":
        ell = 5.0
    if cfg["aniso"]:
        ls = list(np.exp(np.random.default_rng(2).uniform(np.log(0.5), np.log(2.0), size=d)) * ell)
    else:
        ls = ell
    return dict(td=td, n=n, k=k, d=d, R=R, b=b, X=Xd, y=yd, bi=bi, ni=ni, ls=ls,
                mean=torch.empty((b, R), device=dev, dtype=td), var=torch.empty((b,), device=dev, dtype=td),
                info=torch.zeros(1, dtype=torch.int32, device=dev))


def make_step(cfg, w, route: str, path: str, use_packed: bool):
    """The callable one timed step runs.

    route "fused":  muygpys_amd.fused.posterior_mean_var / distributed.sharded_loocv (one library call).
    route "dropin": the call sequence ``integration.install()`` binds into the reference's functor layer
                    (gp/muygps.py:406-551 -> gp/kernels/matern.py:148-168 -> gp/muygps.py:164-259):
                    _crosswise_tensor / _pairwise_tensor -> metric -> / length scale -> kernel fn -> perturb
                    -> train_targets[nn_indices] -> _muygps_posterior_mean + _muygps_diagonal_variance,
                    on lazy handles.  "dropin_plain": the same with the responses in a plain torch tensor,
                    so that the reference's gather materialises (b, k)."""
    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    spec = KernelSpec(cfg["kernel"], cfg["metric"], w["ls"], cfg["noise"])
    if route == "fused":
        if cfg["objective"]:
            # (what an optimiser's objective asks for: the scalars; mean / var stay in the prepared evaluation's buffers,
            # where _outputs_of reads them for the spot check)
            return lambda: D.sharded_loocv(spec, w["X"], w["y"], w["bi"], w["ni"], loss="lool", presharded=True,
                                           packed=use_packed, return_outputs=False)
        return lambda: posterior_mean_var(spec, w["X"], w["X"], w["bi"], w["ni"], w["y"], out_mean=w["mean"],
                                          out_var=w["var"], info=w["info"], path=path, packed=use_packed)
    import torch

    from muygpys_amd import integration
    from muygpys_amd._src.gp.kernels import hip as K
    from muygpys_amd._src.gp.muygps import hip as M
    from muygpys_amd._src.gp.noise import hip as N
    from muygpys_amd._src.gp.tensors import hip as T
    from muygpys_amd.config import config

    config.state.lazy_tensors = True  # what integration.install() switches on
    kfn = {"rbf": K._rbf_fn, "matern05": K._matern_05_fn, "matern15": K._matern_15_fn, "matern25": K._matern_25_fn,
           "matern_inf": K._matern_inf_fn}[cfg["kernel"]]
    metric = T._l2 if cfg["metric"] == "l2" else T._F2
    ytab = integration.table(w["y"]) if route == "dropin" else w["y"].as_subclass(torch.Tensor)
    ls = w["ls"]

    def deform(diffs):
        if cfg["aniso"]:  # Anisotropy.__call__: metric(diffs / length_scales)
            return metric(diffs / torch.as_tensor(ls, device=w["X"].device, dtype=w["td"]))
        dist = metric(diffs)  # Isotropy.__call__: metric(diffs) / l (l2) or / l^2 (F2)
        return dist / (ls if cfg["metric"] == "l2" else ls**2)

    def step():
        # the not-positive-definite counter of a launch is read when the next evaluation arrives instead of right
        # behind the launch (config.py: one device synchronisation per evaluation otherwise); time_steps flushes the
        # last one.  Set per step and put back: nothing of it outlives the loop
        spd_mode, config.state.check_spd = config.state.check_spd, "deferred"
        try:
            cross = T._crosswise_tensor(w["X"], w["X"], w["bi"], w["ni"])
            pair = T._pairwise_tensor(w["X"], w["ni"])
            y_nn = ytab[w["ni"]]
            Kcross, Kin = kfn(deform(cross)), kfn(deform(pair))
            Kin = N._homoscedastic_perturb(Kin, cfg["noise"])
            mean = M._muygps_posterior_mean(Kin, Kcross, y_nn)
            var = M._muygps_diagonal_variance(Kin, Kcross, 1.0)
        finally:
            config.state.check_spd = spd_mode
        return mean, var

    return step


def time_steps(step, warmup: int, steps: int, dist, backend: str, dev, settle_steps: int = 0, event_stride: int = 1):
    """W untimed steps, then K steps bracketed by barrier + synchronize; MAX over ranks.  Per-step HIP
    events on the launch stream (torch's current stream is the one every library call is given)."""
    import torch

    import gc

    # (the interpreter's cyclic collector stays out of the timed region, as in timeit: a full collection of a process
    # with torch and numpy loaded takes ~75 ms -- one hit one step of a 60-step loop of 0.23 ms evaluations and made
    # the line read 81 instead of 530 M/s.  Collected BEFORE the warm-up: 75 ms of idle GPU in front of the timed
    # steps would put them back on the clock ramp)
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    last = None
    # (event_stride: an event every so many steps -- recording one costs ~2 us of host time, 1 % of a 0.2 ms
    # synchronous step; kern_ms then holds the per-step average of each stride)
    event_stride = max(1, min(int(event_stride), steps))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps // event_stride + 1)]
    # (settle_steps: untimed steps that bring the clocks up -- HERE, behind the collection above: run before it, its
    # 75 ms of idle GPU put a short loop, --steps 20 --warmup 5, back on the ramp: 0.422 instead of 0.434)
    for _ in range(settle_steps + warmup):
        last = step()
    torch.cuda.synchronize()
    try:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            last = step()
            if (i + 1) % event_stride == 0:
                ev[(i + 1) // event_stride].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        if gc_was:
            gc.enable()
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) / event_stride for i in range(steps // event_stride)]
    from muygpys_amd import _lib

    _lib.flush_spd_checks()  # (drop-in routes: the last step's not-positive-definite counter, see make_step)
    ranks_seen = 1
    if dist is not None:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=cdev, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # every rank took part in a collective
        ranks_seen = int(ones.item())
    return elapsed, kern_ms, ranks_seen, last


def spot_check(cfg, w, mean, var, rows: int = 256):
    """Largest error of `rows` evenly spaced neighbourhoods of the outputs a timed loop left behind, against the
    fp64 LDS workgroup kernel on plain tables (mgp_posterior_generic_f64: another kernel family, the one every
    register kernel is tested against, itself pinned to the oracle and the reference fixtures in tests/).  Error
    measure and tolerance are the test-suite's (tests/util.py: |a - b| <= rtol (|b| + rms(b)), north_star rtol)."""
    import torch

    from muygpys_amd.fused import KernelSpec, posterior_mean_var

    b, k, R = w["b"], w["k"], w["R"]
    if b == 0:
        return None
    pick = torch.unique(torch.linspace(0, b - 1, min(rows, b), device=w["bi"].device).round().long())
    bi, ni = w["bi"][pick], w["ni"][pick]
    # a compact fp64 copy of just the rows these neighbourhoods touch (the full tables can be gigabytes)
    used = torch.unique(torch.cat([bi, ni.reshape(-1)]))
    X64 = w["X"][used].double()
    y64 = w["y"][used].double()
    bi_c, ni_c = torch.searchsorted(used, bi), torch.searchsorted(used, ni.reshape(-1)).reshape(ni.shape)
    ls = w["ls"]
    spec = KernelSpec(cfg["kernel"], cfg["metric"], ls, cfg["noise"])
    m_ref, v_ref = posterior_mean_var(spec, X64, X64, bi_c, ni_c, y64, path="generic", packed=False)
    m_got = mean.reshape(b, -1)[pick].double().reshape(m_ref.shape)
    v_got = var.reshape(b)[pick].double()

    def rel(got, ref):
        rms = torch.sqrt(torch.mean(ref * ref))
        return float(((got - ref).abs() / (ref.abs() + rms)).max())

    err = max(rel(m_got, m_ref), rel(v_got, v_ref))
    tol = 1e-3 if cfg["dtype"] == "f32" else 1e-5
    return {"rows": int(pick.numel()), "max_rel_err": err, "tol": tol, "ok": bool(err <= tol),
            "against": "mgp_posterior_generic_f64 (fp64 LDS workgroup kernel, plain tables) on the same rows"}


def _outputs_of(cfg, w, route: str, last):
    """(mean, var) device tensors of the last timed step."""
    if route == "fused" and cfg["objective"]:
        from muygpys_amd import distributed as D

        plan = D.last_plan()  # the prepared evaluation the timed steps ran: its buffers hold the last step's outputs
        assert plan is not None and plan.b == w["b"]
        return plan.mean, plan.var
    if route == "fused":
        return w["mean"], w["var"]
    return last[0], last[1]


def _launch_geometry():
    from muygpys_amd import _lib

    return _lib.last_launch_geometry()


def roofline_of(cfg, w, avg_ms: float, kernel_name: str, objective: bool):
    s = 4 if cfg["dtype"] == "f32" else 8
    k, d, R, b = w["k"], w["d"], w["R"], w["b"]
    B = algorithmic_bytes(k, d, R, s, loocv=objective)
    F = algorithmic_flops(k, d, R, loocv=objective)
    achieved = B * b / (avg_ms * 1e-3) / 1e9
    tflops = F * b / (avg_ms * 1e-3) / 1e12
    vpeak = FP32_VECTOR_TFLOPS if cfg["dtype"] == "f32" else FP64_VECTOR_TFLOPS
    traffic, source = measured_traffic(b, k, d, cfg["dtype"])
    # what the memory system actually moved per second (PMC bytes of the committed profile over this run's kernel time):
    # gathered rows come in whole 128-byte lines, so this sits above `achieved` -- the headline kernel's is ~0.65 of peak
    traffic_rate = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
        "traffic_rate_GBps": traffic_rate, "traffic_frac": traffic_rate / HBM_PEAK_GBS if traffic_rate else None,
        "kernel": kernel_name,
        "launch": dict(zip(("workgroups", "lds_bytes_per_workgroup"), _launch_geometry())),
        "algorithmic_bytes_per_neighbourhood": B, "algorithmic_bytes_per_launch": B * b,
        "kernel_ms": avg_ms,
        "valu": {"achieved": tflops, "peak": vpeak, "unit": "TFLOP/s", "frac": tflops / vpeak,
                 "algorithmic_flops_per_neighbourhood": F},
    }


def secondary_line(name: str, cfg_id: int, route: str, args, dev, steps: int = 5, knn: bool = False, **override):
    """A short run of another BASELINE config (or of the drop-in route on the headline config) for
    the driver's one line: value, ms_per_step, roofline / VALU fraction, kernel."""
    import gc

    import torch

    from muygpys_amd import _lib
    from muygpys_amd.fused import PackedTable, clear_caches, pack_table

    cfg = dict(CONFIGS[cfg_id], **override)
    w = build_workload(cfg, dev, 0, knn)
    gathered_route = route == "dropin_plain"
    use_packed = PackedTable.supported(w["d"], w["R"], w["k"], w["td"])
    if use_packed and route == "fused":
        pack_table(w["X"], w["y"])
    step = make_step(cfg, w, route, "auto", use_packed if route == "fused" else "auto")
    grad = None
    if cfg.get("grad"):
        # one evaluation of the LOOCV objective AND its analytic gradient with respect to the length scales (what
        # L_BFGS_B_optimize(..., analytic_gradient=True) runs per iteration: fused.loocv_value_and_grad = the forward
        # launch, the backward launch on the same kernel family, two column sums)
        from muygpys_amd.fused import KernelSpec, loocv_value_and_grad

        gspec = KernelSpec(cfg["kernel"], cfg["metric"], w["ls"], cfg["noise"])
        grad = {}

        def step():  # noqa: F811
            grad["value"], grad["g_ls"], grad["g_noise"] = loocv_value_and_grad(gspec, w["X"], w["y"], w["bi"], w["ni"], loss="lool")
            return {"lool": grad["value"], "sigma_sq": 1.0}
    bwd = None
    if cfg.get("bwd"):
        # the full backward of one prediction batch (what the torch layer's .backward() launches: reference
        # torch/muygps_layer.py:129-164 under autograd): cotangents of the features (scatter-add into the table's
        # gradient), the length scale, the noise diagonal and the responses, from random upstream cotangents
        from muygpys_amd.fused import _length_scale_tensor

        g_ = torch.Generator(device=dev)
        g_.manual_seed(11)
        P = _lib.ptr
        tg = w["y"].reshape(w["n"], -1).contiguous()
        lst = _length_scale_tensor(w["ls"], w["d"], w["X"])
        bwd = dict(gm=torch.randn((w["b"], 1), device=dev, dtype=w["td"], generator=g_),
                   gv=torch.randn((w["b"],), device=dev, dtype=w["td"], generator=g_),
                   gx=torch.zeros_like(w["X"]), gy=torch.zeros_like(tg), lst=lst, tg=tg,
                   gl=torch.empty((w["b"], lst.numel()), device=dev, dtype=w["td"]),
                   gn=torch.empty((w["b"], w["k"]), device=dev, dtype=w["td"]))
        kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[cfg["kernel"]]
        mid = 0 if cfg["metric"] == "l2" else 1

        def run_backward(X, tgt, bi, ni, lst_, gm, gv, gx, gy, gl, gn, info):
            td_ = X.dtype
            rc = _lib.fn("posterior_backward", td_)(P(X), P(X), w["d"], P(bi), P(ni), bi.shape[0], w["k"], P(tgt), 1, 0,
                                                    float(cfg["noise"]), None, kid, mid, P(lst_), lst_.numel(), P(gm), P(gv),
                                                    P(gx), P(gx), P(gy), P(gl), P(gn), P(info), _lib.stream_ptr())
            _lib.check(rc, "mgp_posterior_backward")

        def step():  # noqa: F811
            run_backward(w["X"], bwd["tg"], w["bi"], w["ni"], bwd["lst"], bwd["gm"], bwd["gv"], bwd["gx"], bwd["gy"], bwd["gl"],
                         bwd["gn"], w["info"])
            return None
    acquire = None
    if cfg.get("acquire"):
        # one trial of the Bayes loop = one evaluation + one acquisition step (surrogate fit on the trials so far,
        # 10 000 candidates on the device, polish; _src/optimize/chassis/hip.py) -- 16 recorded trials of this model's
        # d length scales, the middle of the reference's default budget of 26
        from muygpys_amd._src.optimize.chassis.hip import _UCBBayesOpt

        evaluate = step
        rng = np.random.RandomState(5)
        opt = _UCBBayesOpt(lambda **kw: 0.0, [f"length_scale{i}" for i in range(w["d"])],
                           np.array([[0.1, 10.0]] * w["d"]), random_state=7)
        for x in opt._sample(16):
            opt.X.append(x)
            opt.y.append(-float(((np.log(x) - 0.3) ** 2).sum()) + 0.01 * rng.randn())
        acquire = []  # host seconds of every acquisition step (the first ones pay one-time library initialisation)

        def step():  # noqa: F811
            out = evaluate()
            t0 = time.perf_counter()
            opt._suggest(2.576)
            acquire.append(time.perf_counter() - t0)
            return out
    # the first launches after a pause run on ramping clocks (the headline kernel: 2.1 ms falling to 1.5 over ~20
    # launches): warm for >= 60 ms of this config's steps, then time >= 150 ms of them
    # (the estimate of a step: the FASTEST of three after an untimed first one -- one-time costs of the first call, or a
    # cyclic garbage collection falling into a timed pair, made a 0.22 ms step look like 30 ms and the line was then
    # measured over 5 steps on the clock ramp: 498 instead of 570 M/s in one run of round 5)
    step()
    torch.cuda.synchronize()
    est = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        est = min(est, time.perf_counter() - t0)
    est = max(est, 1e-5)
    # (caps of 60 / 40 steps until round 4: a 0.23 ms step was then warmed for 10 ms and timed for 14 -- on the ramp)
    steps = int(min(1000, max(steps, np.ceil(0.15 / est))))
    elapsed, kern_ms, _, last = time_steps(step, int(min(400, max(2, np.ceil(0.06 / est)))), steps, None, "", dev,
                                           event_stride=int(max(1, 0.002 / est)))  # (events >= 2 ms apart)
    if route == "fused" and cfg["objective"]:
        assert np.isfinite(last["lool"]) and np.isfinite(last["sigma_sq"]), f"{name}: non-finite objective"
    avg_ms = float(np.mean(kern_ms))
    if os.environ.get("BENCH_DEBUG"):
        from muygpys_amd import distributed as D_

        print(f"[debug] {name}: est {est * 1e3:.3f} ms, steps {steps}, plans {[(p.b, p.host_result) for p in D_._PLANS.values()]}, "
              f"mean {np.mean(kern_ms):.3f} kern_ms {[round(v, 3) for v in kern_ms][:12]}", file=sys.stderr)
    roof = roofline_of(cfg, w, avg_ms, _lib.last_kernel(), cfg["objective"])  # the instantiation actually launched
    if bwd is not None:
        # bytes of the backward: the forward's, the gradient rows of the k + 1 gathered rows read and written back, the
        # noise and length-scale partials
        es = 4 if cfg["dtype"] == "f32" else 8
        Bb = (algorithmic_bytes(w["k"], w["d"], w["R"], es) + 2 * (w["k"] + 1) * w["d"] * es + w["k"] * es
              + bwd["lst"].numel() * es)
        roof["achieved"] = Bb * w["b"] / (avg_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["traffic"], roof["traffic_source"] = None, None
        # the same cotangents of the first 2 000 neighbourhoods in fp64 (another instantiation, or the LDS workgroup
        # kernel): feature-gradient table and the summed length-scale / noise partials
        m = min(2000, w["b"])
        f64 = torch.float64
        X64, tg64 = w["X"].to(f64), bwd["tg"].to(f64)
        bi_s, ni_s = w["bi"][:m].contiguous(), w["ni"][:m].contiguous()

        def subset(td_, X, tgt):
            gx, gy = torch.zeros_like(X), torch.zeros_like(tgt)
            gl = torch.zeros((m, bwd["lst"].numel()), device=dev, dtype=td_)
            gn = torch.zeros((m, w["k"]), device=dev, dtype=td_)
            info = torch.zeros(1, dtype=torch.int32, device=dev)
            run_backward(X, tgt, bi_s, ni_s, bwd["lst"].to(td_), bwd["gm"][:m].to(td_).contiguous(),
                         bwd["gv"][:m].to(td_).contiguous(), gx, gy, gl, gn, info)
            torch.cuda.synchronize()
            return gx.double(), gy.double(), gl.double().sum(0), gn.double().sum()

        got, ref = subset(w["td"], w["X"], bwd["tg"]), subset(f64, X64, tg64)
        step()  # (`kernel` below: the timed instantiation again)

        def rel(a_, b_):
            return float((a_ - b_).abs().max() / (b_.abs().max() + 1e-300))

        err = max(rel(a_, b_) for a_, b_ in zip(got, ref))
        tol = 2e-2 if cfg["dtype"] == "f32" else 1e-8
        check = {"rows": int(m), "max_rel_err": err, "tol": tol, "ok": bool(err <= tol),
                 "against": "the same cotangents of the first neighbourhoods computed in fp64 (max-norm relative)"}
    elif grad is not None:
        # the gradient against central differences of the objective itself along the first length scale
        from muygpys_amd import distributed as D_
        from muygpys_amd.fused import KernelSpec as KS_

        def lool_at(delta):
            ls = list(w["ls"]) if isinstance(w["ls"], (list, tuple)) else [w["ls"]]
            ls[0] = ls[0] + delta
            sp = KS_(cfg["kernel"], cfg["metric"], ls if len(ls) > 1 else ls[0], cfg["noise"])
            return D_.sharded_loocv(sp, w["X"], w["y"], w["bi"], w["ni"], loss="lool", presharded=True, return_outputs=False)["lool"]

        h = 1e-5 * float(w["ls"][0] if isinstance(w["ls"], (list, tuple)) else w["ls"])
        fd = (lool_at(h) - lool_at(-h)) / (2 * h)
        err = abs(float(grad["g_ls"][0]) - fd) / max(abs(fd), 1e-300)
        tol = 1e-4 if cfg["dtype"] == "f64" else 5e-2
        check = {"rows": int(w["b"]), "max_rel_err": err, "tol": tol, "ok": bool(err <= tol),
                 "against": "central differences of the LOOCV objective along the first length scale"}
    else:
        check = spot_check(cfg, w, *_outputs_of(cfg, w, route, last))
    out = {
        "baseline_config": cfg_id, "route": route, "what": cfg["what"], "dtype": cfg["dtype"],
        "value": w["b"] * steps / elapsed, "unit": "neighborhoods/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "batch": w["b"], "points": w["n"], "nn_count": w["k"], "feature_count": w["d"], "response_count": w["R"],
        "roofline": {"frac": roof["frac"], "achieved": roof["achieved"], "traffic": roof["traffic"],
                     "traffic_source": roof["traffic_source"], "kernel_ms": avg_ms},
        "valu": {"frac": roof["valu"]["frac"], "achieved": roof["valu"]["achieved"], "unit": "TFLOP/s"},
        "kernel": roof["kernel"], "check": check,
    }
    if bwd is not None:
        out["kernel"] = _lib.last_kernel()
        out["note"] = ("the full backward of one prediction batch: feature (scatter-add), length-scale, noise and "
                       "response cotangents in one launch; roofline bytes = the forward's + the gathered rows' gradient "
                       "read and written back")
    if grad is not None:
        out["note"] = ("one LOOCV objective evaluation AND its analytic gradient with respect to the length scales "
                       "(fused.loocv_value_and_grad); `kernel` names the backward launch")
    if acquire is not None:
        timed = acquire[-steps:]  # (the timed steps' acquisitions: warm-up ones carry one-time library initialisation)
        out["acquisition_ms"] = float(np.median(timed)) * 1e3
        out["acquisition_ms_mean"] = float(np.mean(timed)) * 1e3
        out["note"] = ("one trial of the Bayes loop: the LOOCV evaluation of this shard plus one acquisition step "
                       "(host clock around _UCBBayesOpt._suggest, 16 recorded trials; median / mean over the timed steps)")
    if override or knn:
        out["override"] = dict(override, **({"neighbours": "exact kNN (GPU brute force)"} if knn else {}))
    if gathered_route:
        out["note"] = ("responses in a plain torch tensor: the reference's own train_targets[nn_indices] materialises "
                       "(b, k) per step (a torch gather inside the timed step); features from the prepared table")
    del w, step, last
    gc.collect()
    clear_caches()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: steady state -- the first ~20 launches after an idle GPU run on ramping clocks, up to 30 % slower;
    # `--steps 20 --warmup 3` is the convention of rounds 1-2 and lands inside the ramp)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="run the step for this long before the W warm-ups (clock ramp); 0: the rounds-1-4 convention")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config")
    ap.add_argument("--route", default="fused", choices=["fused", "dropin", "dropin_plain"],
                    help="fused: one library call per step; dropin: the family-level call sequence "
                         "integration.install() binds (lazy handles); dropin_plain: the same, plain response tensor")
    ap.add_argument("--points", type=int, default=0, help="training points N (replicated per GPU; 0 = the config's)")
    ap.add_argument("--batch", type=int, default=0, help="neighbourhoods per GPU per step (0 = the config's)")
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--d", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "f32", "f64"])
    ap.add_argument("--kernel", default="")
    ap.add_argument("--objective", action="store_true",
                    help="time one LOOCV objective evaluation (fused launch + loss sums + the all-reduce)")
    ap.add_argument("--knn", action="store_true", help="exact GPU kNN neighbourhoods instead of random rows")
    ap.add_argument("--cpu-sample", type=int, default=CPU_BATCH,
                    help="neighbourhoods of the CPU baseline leg (default: BASELINE.md sec. 2's 65 536); 0 disables it")
    ap.add_argument("--path", default="auto", choices=["auto", "generic", "rhs"], help="kernel family to time")
    ap.add_argument("--no-prepared-tables", action="store_true", help="read the plain feature / target tables")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of configs 3 / 4 / 5 and of the drop-in route attached as `secondary`")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own batch of the config's size; strong: ONE global batch of that size "
                         "sharded over the ranks with the reference's chunk rule (BASELINE config 3)")
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

    from muygpys_amd import _lib
    from muygpys_amd.fused import PackedTable, pack_table

    cfg = dict(CONFIGS[args.config])
    for key, val in (("points", args.points), ("batch", args.batch), ("k", args.k), ("d", args.d),
                     ("dtype", args.dtype), ("kernel", args.kernel)):
        if val:
            cfg[key] = val
    if args.objective:
        cfg["objective"] = True
    if args.route != "fused" and cfg["objective"]:
        raise SystemExit("bench.py: --route dropin times the prediction call sequence (configs 2 / 5)")
    strong = args.scaling == "strong"
    w = build_workload(cfg, dev, rank, args.knn, world, strong)
    td, n, k, d, R, b = w["td"], w["n"], w["k"], w["d"], w["R"], w["b"]
    total_b = min(cfg["batch"], n) if strong else world * b  # neighbourhoods all ranks process per step
    # the prepared tables are built once, outside the timed loop (they are constant across all
    # objective evaluations / prediction batches of a model; DESIGN.md sec. 5 gives the pack time)
    use_packed = (not args.no_prepared_tables) and args.path == "auto" and PackedTable.supported(d, R, k, td)
    pack_ms = pack_cold_ms = None
    if use_packed:
        pack_table(w["X"], w["y"])  # the cached table the timed steps use
        # one more pack, timed on the stream: what a pack per step would add.  Twice: the first one also pays the
        # allocator's hipMalloc of a block this size (`cold`, what round 4 reported: 0.36-0.45 ms around a 60 us
        # kernel); a pack per step would find the block in torch's caching allocator, as the second one does
        pack_cold_ms = None
        for attempt in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            spare = PackedTable(w["X"], w["y"])
            e1.record()
            torch.cuda.synchronize()
            pack_ms = e0.elapsed_time(e1)
            if attempt == 0:
                pack_cold_ms = pack_ms
            del spare
    step = make_step(cfg, w, args.route, args.path, use_packed if args.route == "fused" else "auto")
    # Two conventions, both reported (VERDICT r04 #1c).  `ramp`: W + K steps from an idle GPU -- what rounds 1-4 put in
    # `value`; with the driver's short loops (--steps 20 --warmup 5 = 40 ms) it lies inside the clock ramp (the kernel's
    # launches read 2.1, 2.0, 1.85 ... 1.5 ms over the first ~25).  `value`: the same W + K steps after the GPU has run
    # the step for --settle-ms (150 ms: what the secondary lines have always done) -- steady state.
    ramp, settle_steps = None, 0
    if args.settle_ms > 0:
        r_elapsed, r_kern, _, _ = time_steps(step, args.warmup, args.steps, dist, args.backend, dev)
        ramp = {"value": total_b * args.steps / r_elapsed, "ms_per_step": r_elapsed / args.steps * 1e3,
                "kernel_ms": float(np.mean(r_kern)),
                "what": f"{args.warmup} + {args.steps} steps from an idle GPU (the convention of rounds 1-4)"}
        # (a step count, the same on every rank -- r_elapsed is the maximum over the ranks --, not a clock: a step of an
        # objective config ends in a collective)
        settle_steps = int(min(2000, np.ceil(args.settle_ms * 1e-3 / max(r_elapsed / args.steps, 1e-5))))
    elapsed, kern_ms, ranks_seen, last = time_steps(step, args.warmup, args.steps, dist, args.backend, dev,
                                                    settle_steps=settle_steps)
    non_spd = int(w["info"].item())
    kernel_name = _lib.last_kernel()  # the instantiation the timed steps actually launched (this thread's last call)
    if cfg["objective"]:
        assert np.isfinite(last["lool"]) and np.isfinite(last["sigma_sq"]), "non-finite objective"
    elif args.route == "fused":
        assert torch.isfinite(w["mean"]).all() and torch.isfinite(w["var"]).all(), "non-finite outputs"
    else:
        assert torch.isfinite(last[0]).all() and torch.isfinite(last[1]).all(), "non-finite outputs"

    if rank == 0:
        avg_ms = float(np.mean(kern_ms))
        roof = roofline_of(cfg, w, avg_ms, kernel_name, cfg["objective"])
        check = spot_check(cfg, w, *_outputs_of(cfg, w, args.route, last))
        assert check is None or check["ok"], f"timed outputs differ from the fp64 workgroup kernel: {check}"
        if ramp is not None:  # (inside `roofline`: the driver's record keeps it; the top-level `ramp` it drops)
            roof["ramp_value"], roof["ramp_ms_per_step"] = ramp["value"], ramp["ms_per_step"]
            roof["settle_ms"] = args.settle_ms
        if pack_ms is not None:
            roof["prepared_table_pack_kernel_ms"] = pack_ms
            roof["prepared_table_pack_cold_ms"] = pack_cold_ms
            roof["frac_with_pack_per_step"] = roof["frac"] * avg_ms / (avg_ms + pack_ms)
        out = {
            "metric": "neighborhoods/sec (posterior mean+var)" if not cfg["objective"]
                      else "neighborhoods/sec (LOOCV objective evaluation)",
            "value": total_b * args.steps / elapsed,
            "unit": "neighborhoods/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "collective_backend": None if dist is None else ("rccl" if args.backend == "nccl" else args.backend),
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_ms": args.settle_ms,
            "ramp": ramp,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": cfg["dtype"],
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config}: {cfg['what']}; {cfg['kernel']} nu-fixed, {n} synthetic "
                            f"points, d={d}, nn_count={k}, responses={R}, "
                            + (f"ONE batch of {total_b} neighbourhoods per step sharded over {world} GPU(s) ({b} on rank 0), "
                               if strong else f"{b} neighbourhoods per GPU per step, ") +
                            f"{'exact kNN' if args.knn else 'random'} neighbour rows, "
                            f"{'Anisotropy' if cfg['aniso'] else 'Isotropy'}/{cfg['metric']}, noise={cfg['noise']}",
                "baseline_config": args.config, "route": args.route,
                "points": n, "batch_per_gpu": b, "batch_all_gpus": total_b, "nn_count": k, "feature_count": d, "response_count": R,
                "kernel_path": args.path, "prepared_tables": bool(use_packed),
                "prepared_table_pack_ms": pack_ms,
                "non_spd_neighbourhoods": non_spd,
            },
            "roofline": roof,
            "check": check,
        }
        if cfg["objective"]:
            out["config"]["lool"] = last["lool"]
            out["config"]["sigma_sq"] = last["sigma_sq"]
            out["roofline"]["note"] = ("kernel_ms is the whole step (fused launch + fp64 loss reductions + host finish "
                                       "+ all-reduce), so achieved is a lower bound for the fused kernel alone")
        if args.route != "fused":
            out["roofline"]["note"] = ("kernel_ms is the whole family-level call sequence of one prediction batch "
                                       "(handles, one fused launch, its host glue)")
        if world == 1 and not args.no_secondary:
            # the other BASELINE configs and the drop-in route, 5 steps each (the headline keys above are
            # unaffected: these run after the timed region, on their own tables)
            del w, step, last
            from muygpys_amd.fused import clear_caches

            clear_caches()
            torch.cuda.empty_cache()
            sec = {}
            # knn: config 2 on exact k-NN neighbourhoods (real gather locality instead of uniform-random rows);
            # points8M: config 2 on an 8 M-row table (1.5 GB prepared: far beyond the 256 MB Infinity Cache);
            # c3_shard8: one LOOCV evaluation of 125 k neighbourhoods = what each of 8 ranks runs under --scaling strong
            # (c3_shard8 right behind c3, the line it is divided by: as the last line, behind the 1.5 GB table of points8M
            # and its empty_cache(), it read 0.219 ms per step in some runs and 0.26 in others -- tools/c3bench.py in a
            # fresh process is stable at 0.222-0.227 --, i.e. it measured where the allocator had put its table)
            plan = [("dropin", 2, "dropin", {}), ("dropin_plain", 2, "dropin_plain", {}), ("c3", 3, "fused", {}),
                    ("c3_shard8", 3, "fused", {"batch": 125_000}), ("c2_bwd", 2, "fused", {"bwd": True}),
                    ("c4", 4, "fused", {}), ("c4_shard8", 4, "fused", {"batch": 1_250_000, "acquire": True}),
                    ("c4_grad", 4, "fused", {"batch": 2_000_000, "grad": True}),
                    ("c5", 5, "fused", {}), ("knn", 2, "fused", {"knn": True}),
                    ("points8M", 2, "fused", {"points": 8_000_000})]
            for name, cid, route, over in plan:
                if cid == args.config and route == args.route and not over:
                    continue
                try:
                    sec[name] = secondary_line(name, cid, route, args, dev, **over)
                except Exception as exc:  # a secondary line must never take the headline down
                    sec[name] = {"error": f"{type(exc).__name__}: {exc}"}
            if "value" in sec.get("c3_shard8", {}) and "value" in sec.get("c3", {}):
                # what strong scaling over 8 GPUs can reach at best: the 125 k-neighbourhood launch's rate relative to
                # the 1 M-neighbourhood launch's (grid fill, clock ramp and the fixed cost of an evaluation)
                sec["c3_shard8"]["rate_vs_full_batch"] = sec["c3_shard8"]["value"] / sec["c3"]["value"]
            if "value" in sec.get("c4_shard8", {}) and "value" in sec.get("c4", {}):
                sec["c4_shard8"]["rate_vs_full_batch"] = sec["c4_shard8"]["value"] / sec["c4"]["value"]
            out["secondary"] = sec
            # the same, compact, inside `roofline` (the driver's record keeps `roofline` and `config` whole and drops
            # unknown top-level keys): value [neighbourhoods/s], ms per step, fraction of the HBM roofline / of the
            # vector peak, kernel
            out["roofline"]["secondaries"] = {
                name: ({"error": v["error"]} if "error" in v else dict(
                    {"value": v["value"], "ms_per_step": v["ms_per_step"], "frac": v["roofline"]["frac"],
                     "valu_frac": v["valu"]["frac"], "kernel": v["kernel"], "batch": v["batch"],
                     "check_ok": None if v["check"] is None else v["check"]["ok"]},
                    **{key: v[key] for key in ("rate_vs_full_batch", "acquisition_ms", "acquisition_ms_mean") if key in v}))
                for name, v in sec.items()}
        if args.cpu_sample > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(k, d, args.cpu_sample, 20241008, points=min(n, 1_000_000))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
