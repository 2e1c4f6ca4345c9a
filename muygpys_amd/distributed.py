"""Multi-GPU execution of the hot path: one process per GPU, batch rows sharded, tables replicated.

The reference's only parallelism is data parallelism over the batch (neighbourhood) dimension:
its ``mpi`` backend gives rank r a contiguous block of rows -- ``floor(b / P)`` each, the
remainder going to the LAST ``b mod P`` ranks (src/MuyGPyS/_src/mpi_utils.py:36-41) -- and
reduces scalars with ``allreduce(SUM)`` (``_src/optimize/scale/mpi.py:35-36``,
``_src/optimize/loss/mpi.py:23-24,57``).  Rank 0 builds every tensor and scatters chunks
there (mpi_utils.py:56-96); here nothing is scattered: every rank holds the (small) feature
and target tables and gathers its own rows, so no tensor ever crosses xGMI.

Prediction needs no collective at all.  One LOOCV objective evaluation needs ONE all-reduce
of ``4 + R`` float64 scalars (the reference uses three), because with
``v_i`` the unscaled variances and ``r_i`` the residuals

    sigma^2        = sum_i y_i^T K_i^-1 y_i / (n k)
    lool(sigma^2)  = (1 / sigma^2) sum r_i^2 / v_i + sum log v_i + n log sigma^2
    mse            = sum r_i^2 / n

are all functions of the partial sums ``[sum r^2/v, sum log v, sum r^2, n, sum y^T K^-1 y]``.
``looph`` is not separable in sigma^2 and takes a second pass + a one-scalar all-reduce;
``pseudo_huber`` is one more entry of the same vector.

Two routes reach the optimisers: :func:`spec_objective` (an ``obj_fn`` straight on
:func:`sharded_loocv`, one all-reduce per evaluation) and :class:`sharded_reductions` /
:func:`optimize_sharded` (the reference's mpi-backend layout: the functor layer is unchanged and
the hip loss / scale functions all-reduce their sums, two collectives per evaluation).

``torch.distributed`` with backend ``nccl`` is RCCL on ROCm; the payload (<= 160 bytes) is
pure latency over xGMI.  The CPU test-suite drives the same code over ``gloo``.
"""

from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import torch


def chunk_sizes(count: int, size: int) -> List[int]:
    """Rows per rank, reference rule (mpi_utils.py:36-41): the remainder goes to the last ranks."""
    floor = int(count / size)
    remainder = count - floor * size
    return [floor + 1 if i >= (size - remainder) else floor for i in range(size)]


def shard_bounds(count: int, rank: int, size: int):
    sizes = chunk_sizes(count, size)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def shard_rows(x, rank: int, size: int):
    """This rank's contiguous block of rows of a batch-leading tensor."""
    lo, hi = shard_bounds(x.shape[0], rank, size)
    return x[lo:hi]


def _world(group=None):
    dist = _dist()
    if dist is not None and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_DIST = False


def _dist():
    """torch.distributed, or None when this torch has none (imported once: the functions below sit in front of every
    objective evaluation)."""
    global _DIST
    if _DIST is False:
        import torch.distributed as dist

        _DIST = dist if dist.is_available() else None
    return _DIST


def _collectives_on(group=None) -> bool:
    """A process group of more than one rank -- or of ONE rank with ``MUYGPYS_HIP_FORCE_COLLECTIVES=1``: a single-GPU
    box then runs every collective of the path through torch's RCCL (``nccl``) for real, on device tensors
    (tests/test_gpu_distributed.py), instead of short-circuiting them."""
    import os

    dist = _dist()
    if dist is None or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("MUYGPYS_HIP_FORCE_COLLECTIVES") == "1"


def allreduce_sum_(partials: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the partial-sum vector (RCCL over xGMI on GPUs; gloo on CPU)."""
    import torch.distributed as dist

    if _collectives_on(group):
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    return partials


def gather_rows(local: torch.Tensor, sizes: Sequence[int], group=None) -> torch.Tensor:
    """Every rank's block of rows, concatenated in rank order, on every rank (the reference's test-time
    ``_consistent_unchunk_tensor``, _src/mpi_utils.py:118-141: a pickled allgather of unequal chunks).  The blocks of
    the reference's chunk rule differ by one row when the batch does not divide: they are padded to the longest and
    gathered as ONE equal-sized ``all_gather_into_tensor`` (RCCL's all-gather takes equal counts), then trimmed."""
    import torch.distributed as dist

    size, longest = len(sizes), max(sizes)
    padded = local.new_zeros((longest,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    out = local.new_empty((size * longest,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return torch.cat([out[r * longest: r * longest + sizes[r]] for r in range(size)])


# ---------------------------------------------------------------------------------------------
# Sharded-reduction mode: the hip backend's loss / scale functions all-reduce their sums, like the
# reference's mpi backend does inside _mse_fn / _lool_fn / _looph_fn / _pseudo_huber_fn
# (_src/optimize/loss/mpi.py:20-104) and _analytic_scale_optim (_src/optimize/scale/mpi.py:16-37).
# With it on, the unchanged functor layer (OptimizeFn, make_loo_crossval_fn, MuyGPS.optimize_scale)
# evaluates the GLOBAL objective from every rank's shard of the batch, and every rank sees the same
# scalar -- so deterministic drivers (L-BFGS-B) walk the same trajectory on every rank.
# ---------------------------------------------------------------------------------------------
_ACTIVE = {"on": False, "group": None}


class sharded_reductions:
    """Context manager: ``with sharded_reductions(group): L_BFGS_B_optimize(...)``."""

    def __init__(self, group=None):
        self.group = group

    def __enter__(self):
        self._saved = dict(_ACTIVE)
        _ACTIVE.update(on=True, group=self.group)
        return self

    def __exit__(self, *exc):
        _ACTIVE.update(self._saved)
        return False


def enable_sharded_mode(group=None) -> None:
    """Process-wide switch, the equivalent of running the reference with ``MUYGPYS_BACKEND=mpi`` (its
    ``_is_mpi_mode()`` is process-global): from here on the loss / scale functions all-reduce their sums and a
    ``Parameter("sample" | "log_sample")`` is rank 0's draw on every rank AT CONSTRUCTION
    (gp/hyperparameter/scalar.py:145-146) -- call it on every rank right after ``init_process_group``, before any
    model is built.  (Models built outside a sharded block keep per-rank draws; the optimisation drivers then start
    from rank 0's values anyway, see ``_get_opt_lists`` in ``_src/optimize/chassis/hip.py``.)"""
    _ACTIVE.update(on=True, group=group)


def disable_sharded_mode() -> None:
    _ACTIVE.update(on=False, group=None)


def reductions_active() -> bool:
    return bool(_ACTIVE["on"]) and _collectives_on(_ACTIVE["group"])


def active_group():
    """The process group of the enclosing :class:`sharded_reductions` block (None: the default group)."""
    return _ACTIVE["group"]


def reduce_if_sharded_(sums: torch.Tensor) -> torch.Tensor:
    """All-reduce a vector of partial sums when the sharded-reduction mode is on (else a no-op)."""
    if reductions_active():
        allreduce_sum_(sums, _ACTIVE["group"])
    return sums


def broadcast_scalar(value: float, group=None, root: int = 0) -> float:
    """Rank ``root``'s value on every rank (reference: comm_world.bcast for sampled hyper-parameters,
    gp/hyperparameter/scalar.py:145-146)."""
    import torch.distributed as dist

    if not _collectives_on(group):
        return float(value)
    # nccl (= RCCL) needs a device tensor: this rank's current device (ranks set it from LOCAL_RANK)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.broadcast(t, src=root, group=group)
    return float(t.item())


def broadcast_vector(values, group=None, root: int = 0):
    """Rank ``root``'s float64 vector on every rank (the start point of a sharded optimisation)."""
    import numpy as np
    import torch.distributed as dist

    values = np.asarray(values, dtype=np.float64)
    if not _collectives_on(group):
        return values
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.as_tensor(values, dtype=torch.float64).to(dev)
    dist.broadcast(t, src=root, group=group)
    return t.cpu().numpy()


def synchronized_seed(group=None) -> int:
    """One random seed for all ranks (drawn on rank 0): the stochastic drivers (Bayes-opt) then
    propose the same points everywhere."""
    import numpy as np

    return int(broadcast_scalar(float(np.random.randint(0, 2**31 - 1)), group))


# partial-sum vector layout
P_R2_OVER_V, P_LOG_V, P_R2, P_COUNT, P_HUBER, P_YKY0 = 0, 1, 2, 3, 4, 5
LOSSES = ("lool", "mse", "looph", "pseudo_huber")


# Prepared evaluations (fused.LoocvPlan) of the optimisers' inner loop: one per (tables, shard, model structure).  The
# tensors are referenced by the plan, so their ids stay theirs while the entry lives.
_PLANS: "Dict[tuple, object]" = {}
_PLANS_MAX = 8


def _loocv_plan(spec, features, targets, batch_indices, nn_indices, packed, huber_delta: float, host_result: bool):
    from muygpys_amd.fused import LoocvPlan

    noise_t = spec.noise if isinstance(spec.noise, torch.Tensor) and spec.noise.ndim >= 1 else None
    ls = spec.length_scale
    aniso = not isinstance(ls, (int, float)) and not (isinstance(ls, torch.Tensor) and ls.numel() == 1)
    # (an optimiser's loop asks for the same plan thousands of times: one identity check before the dictionary)
    global _LAST_PLAN
    lp = _LAST_PLAN
    if (lp is not None and lp[0] is features and lp[1] is targets and lp[2] is batch_indices and lp[3] is nn_indices
            and lp[4] == (spec.kernel, spec.metric, aniso, packed, huber_delta, host_result) and lp[5] is noise_t
            and lp[6] == _versions(features, targets, batch_indices, nn_indices, noise_t) and lp[7] == _lib_raw_stream()):
        return lp[8]
    # (the plan holds converted copies of indices / noise whenever dtype or layout differ: an in-place update of the
    # caller's tensors must make a new plan, like one of the tables)
    versions = _versions(features, targets, batch_indices, nn_indices, noise_t)
    key = (spec.kernel, spec.metric, aniso, id(features), id(targets), id(batch_indices), id(nn_indices),
           None if noise_t is None else id(noise_t), versions, str(packed), float(huber_delta), bool(host_result),
           int(_lib_raw_stream()))
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) >= _PLANS_MAX:
            _PLANS.pop(next(iter(_PLANS)))
        plan = LoocvPlan(spec.kernel, spec.metric, features, targets, batch_indices, nn_indices, anisotropic=aniso,
                         noise_tensor=noise_t, huber_delta=huber_delta, packed=packed, host_result=host_result)
        _PLANS[key] = plan
    _LAST_PLAN = (features, targets, batch_indices, nn_indices, (spec.kernel, spec.metric, aniso, packed, huber_delta, host_result),
                  noise_t, versions, _lib_raw_stream(), plan)
    return plan


def _versions(*tensors) -> tuple:
    return tuple(-1 if t is None else t._version for t in tensors)


_LAST_PLAN = None


def _lib_raw_stream() -> int:
    from muygpys_amd import _lib

    return _lib.raw_stream()


def last_plan():
    """The prepared evaluation the most recent :func:`hip_local_partials` call ran (None before the first): its
    ``mean`` / ``var`` buffers hold that call's outputs until the next evaluation on the same tables."""
    return None if _LAST_PLAN is None else _LAST_PLAN[8]


def clear_plans() -> None:
    global _LAST_PLAN
    _PLANS.clear()
    _LAST_PLAN = None


def hip_local_partials(spec, features, targets, batch_indices, nn_indices, packed="auto", huber_delta: float = 1.5,
                       host_result: bool = False, own_outputs: bool = True):
    """The local shard's partial sums on the GPU: ONE launch (``mgp_loocv_*``: the fused kernel walks the fixed-order
    fp64 reduction tree itself) through a prepared evaluation (:class:`muygpys_amd.fused.LoocvPlan`: tables, buffers
    and the argument list are set up once per search, an evaluation costs one ctypes call).

    Returns ``(partials float64 [6], mean, var)``: ``partials`` a device tensor, or -- ``host_result``: a single
    process needs no all-reduce -- a numpy array read from the pinned host memory the kernel wrote it to.

    The plan is cached and its buffers are rewritten by the next evaluation on the same tables.  ``own_outputs``
    (default) hands back copies: ``mean`` / ``var`` (and a device ``partials``) stay what this call computed whatever
    is evaluated afterwards.  ``own_outputs=False`` returns ``mean = var = None`` and the plan's own ``partials``
    buffer (to be consumed before the next evaluation) -- what an optimiser's objective, which only needs the scalar,
    asks for."""
    general = spec.kernel == "matern_gen"
    if general:  # (the general-smoothness model is not on the prepared-evaluation path)
        from muygpys_amd.fused import loocv_partials

        res = loocv_partials(spec, features, targets, batch_indices, nn_indices, huber_delta=huber_delta, packed=packed)
        return res if own_outputs else (res[0], None, None)
    plan = _loocv_plan(spec, features, targets, batch_indices, nn_indices, packed, huber_delta, host_result)
    ls = spec.length_scale
    if isinstance(ls, torch.Tensor):
        ls = ls.detach().cpu().tolist() if ls.numel() > 1 else float(ls)
    noise = 0.0 if isinstance(spec.noise, torch.Tensor) and spec.noise.ndim >= 1 else float(spec.noise)
    plan.launch(ls, noise)
    if not own_outputs:
        return (plan.wait() if host_result else plan.partials), None, None
    # (clones are ordered behind the launch on the same stream; wait() returns a fresh numpy array)
    mean, var = plan.mean.clone(), plan.var.clone()
    return (plan.wait() if host_result else plan.partials.clone()), mean, var


def hip_local_looph(mean, targets_b, var, sigma_sq: float, looph_delta: float = 3.0) -> torch.Tensor:
    """Second pass of ``looph`` (not separable in sigma^2): this shard's
    sum 2 d^2 (sqrt(1 + r^2 / (d^2 s v)) - 1) + log(s v) as a 1-element float64 device tensor."""
    from muygpys_amd import _lib

    if mean.numel() == 0:
        return torch.zeros(1, device=mean.device, dtype=torch.float64)
    s = torch.tensor([sigma_sq], device=mean.device, dtype=torch.float64)
    return _lib.loss_sums(mean.contiguous(), targets_b.contiguous(), var, s, 1.5, looph_delta)[3:4].clone()


def finish_objective(partials: Sequence[float], nn_count: int, loss: str = "lool") -> Dict[str, float]:
    """Global sigma^2 and loss from the (all-reduced) partial sums."""
    p = [float(v) for v in partials]
    n = p[P_COUNT]
    sigma_sq = [yk / (n * nn_count) for yk in p[P_YKY0:]]
    s = sigma_sq[0]
    out = {"sigma_sq": s, "sigma_sq_all": sigma_sq, "count": n, "mse": p[P_R2] / n, "pseudo_huber": p[P_HUBER]}
    out["lool"] = p[P_R2_OVER_V] / s + p[P_LOG_V] + n * math.log(s)
    if loss in out:
        out["objective"] = -out[loss]
    return out


def sharded_loocv(
    spec,
    features: torch.Tensor,
    targets: torch.Tensor,
    batch_indices: torch.Tensor,
    nn_indices: torch.Tensor,
    loss: str = "lool",
    group=None,
    presharded: bool = False,
    local_fn: Callable = hip_local_partials,
    looph_fn: Callable = hip_local_looph,
    packed="auto",
    loss_kwargs: Optional[Dict] = None,
    return_outputs: bool = True,
) -> Dict:
    """One LOOCV objective evaluation over all ranks (objective.py:101-103 semantics: the value
    under ``"objective"`` is MINUS the loss).

    ``batch_indices`` / ``nn_indices`` are the global batch unless ``presharded``; each rank
    evaluates its block and the partial sums meet in ONE all-reduce (``lool``, ``mse``,
    ``pseudo_huber``); ``looph`` is not separable in sigma^2 and takes a second, one-scalar
    all-reduce after sigma^2 is known (reference: three all-reduces, loss/mpi.py:57-104 +
    scale/mpi.py:35-36).  Returns the global scalars plus this rank's ``mean`` / ``var`` (which stay
    sharded, like the reference's results): tensors of their own -- a later evaluation does not change them.
    ``return_outputs=False`` leaves them out (``None``), which saves the two copies; ``looph`` needs them and
    keeps them whatever the flag."""
    if loss not in LOSSES:
        raise ValueError(f"sharded_loocv supports {LOSSES}, not {loss!r}")
    loss_kwargs = dict(loss_kwargs or {})
    rank, size = _world(group)
    if not presharded:
        batch_indices = shard_rows(batch_indices, rank, size)
        nn_indices = shard_rows(nn_indices, rank, size)
    kw = {}
    if local_fn is hip_local_partials:
        # (one process: the kernel writes the sums to pinned host memory and nothing is all-reduced)
        kw = dict(packed=packed, huber_delta=float(loss_kwargs.get("boundary_scale", 1.5)), host_result=not _collectives_on(group),
                  own_outputs=return_outputs or loss == "looph")
    partials, mean, var = local_fn(spec, features, targets, batch_indices, nn_indices, **kw)
    if isinstance(partials, torch.Tensor):
        allreduce_sum_(partials, group)
        partials = partials.tolist()
    out = finish_objective(partials, nn_indices.shape[1], loss)
    if loss == "looph":
        part = looph_fn(mean, targets[batch_indices], var, out["sigma_sq"], float(loss_kwargs.get("boundary_scale", 3.0)))
        allreduce_sum_(part, group)
        out["looph"] = float(part[0])
        out["objective"] = -out["looph"]
    out.update(mean=mean, var=var, rank=rank, world_size=size)
    return out


def spec_objective(spec_fn: Callable, features, targets, batch_indices, nn_indices, loss: str = "lool", group=None,
                   presharded: bool = False, **kwargs) -> Callable:
    """``obj_fn(**hyper)`` for the optimisation drivers, built directly on :func:`sharded_loocv`:
    ``spec_fn(**hyper)`` returns the :class:`KernelSpec` of a trial point; every rank evaluates its
    shard, one all-reduce, the same scalar on every rank (reference: _make_mpi_obj_fn,
    _src/optimize/loss/mpi.py:28-34)."""
    rank, size = _world(group)
    if not presharded:
        batch_indices = shard_rows(batch_indices, rank, size).contiguous()
        nn_indices = shard_rows(nn_indices, rank, size).contiguous()

    def obj_fn(**hyper):
        return sharded_loocv(spec_fn(**hyper), features, targets, batch_indices, nn_indices, loss=loss, group=group,
                             presharded=True, return_outputs=False, **kwargs)["objective"]

    return obj_fn


def optimize_sharded(muygps, features, targets, batch_indices, nn_indices, optimizer: str = "lbfgsb", loss_fn=None,
                     group=None, loss_kwargs: Optional[Dict] = None, **opt_kwargs):
    """Hyper-parameter optimisation of a functor-layer model over all ranks: every rank builds the
    (lazy) training tensors of ITS block of the batch (reference chunk rule), and the unchanged
    ``L_BFGS_B_optimize`` / ``Bayes_optimize`` run under :class:`sharded_reductions`, so the loss and
    sigma^2 are global and identical everywhere (reference: the mpi backend's loss / scale functions,
    optimize/chassis.py:23-195 on scattered tensors).  The Bayes driver's random state is drawn on
    rank 0.  Returns the optimised model (the same on every rank)."""
    from muygpys_amd.optimize import Bayes_optimize, L_BFGS_B_optimize
    from muygpys_amd.optimize.loss import lool_fn

    rank, size = _world(group)
    bi = shard_rows(batch_indices, rank, size).contiguous()
    ni = shard_rows(nn_indices, rank, size).contiguous()
    crosswise, pairwise, batch_targets, batch_nn_targets = muygps.make_train_tensors(bi, ni, features, targets)
    driver = {"lbfgsb": L_BFGS_B_optimize, "bayes": Bayes_optimize}[optimizer]
    if optimizer == "bayes" and opt_kwargs.get("random_state") is None:
        opt_kwargs["random_state"] = synchronized_seed(group)
    with sharded_reductions(group):
        return driver(muygps, batch_targets, batch_nn_targets, crosswise, pairwise,
                      loss_fn=loss_fn or lool_fn, loss_kwargs=dict(loss_kwargs or {}), **opt_kwargs)


def sharded_posterior(spec, test_features, train_features, train_targets, batch_indices, nn_indices, group=None,
                      gather: bool = False):
    """Posterior mean / unscaled variance of this rank's block of the batch; no collective
    unless ``gather`` (then every rank receives the concatenated results, in rank order)."""
    from muygpys_amd.fused import posterior_mean_var

    rank, size = _world(group)
    bi = shard_rows(batch_indices, rank, size)
    ni = shard_rows(nn_indices, rank, size)
    mean, var = posterior_mean_var(spec, test_features, train_features, bi, ni, train_targets)
    if gather and _collectives_on(group):
        sizes = chunk_sizes(nn_indices.shape[0], size)
        mean, var = gather_rows(mean, sizes, group), gather_rows(var, sizes, group)
    return mean, var


def sharded_batch_nns(nbrs_lookup, batch_indices: torch.Tensor, group=None, rank: Optional[int] = None,
                      world_size: Optional[int] = None):
    """Neighbour search for this rank's block of ``batch_indices`` (reference rule), with the table
    replicated like every other table of the path: no collective, the ``(rows, nn_count)`` index and
    distance tensors stay sharded exactly like the posterior outputs they feed.  ``nbrs_lookup`` is
    a :class:`muygpys_amd.neighbors.NN_Wrapper` built from the (replicated) training features.
    Returns ``(local_batch_indices, local_nn_indices, local_distances)``."""
    if rank is None or world_size is None:
        rank, world_size = _world(group)
    local = shard_rows(batch_indices, rank, world_size).contiguous()
    nn_indices, distances = nbrs_lookup.get_batch_nns(local)
    return local, nn_indices, distances
