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
``looph`` is not separable in sigma^2 and takes a second pass + all-reduce.

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
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def allreduce_sum_(partials: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the partial-sum vector (RCCL over xGMI on GPUs; gloo on CPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    return partials


# partial-sum vector layout
P_R2_OVER_V, P_LOG_V, P_R2, P_COUNT, P_YKY0 = 0, 1, 2, 3, 4


def hip_local_partials(spec, features, targets, batch_indices, nn_indices):
    """The local shard's partial sums on the GPU: one fused launch + two fp64 reductions.

    Returns ``(partials float64 [4 + R], mean, var)`` -- all device tensors."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import posterior_mean_var

    mean, var, yk = posterior_mean_var(spec, features, features, batch_indices, nn_indices, targets, want_ykinvy=True)
    b = nn_indices.shape[0]
    R = 1 if targets.ndim == 1 else targets.shape[1]
    out = torch.zeros(4 + R, device=features.device, dtype=torch.float64)
    if b > 0:
        if R != 1:
            raise NotImplementedError("the LOOCV losses are defined for a single response (reference: loss/numpy.py:34-61)")
        yb = targets[batch_indices]
        sums = torch.empty(6, device=features.device, dtype=torch.float64)
        _lib.check(
            _lib.fn("loss_sums", mean.dtype)(
                _lib.ptr(mean.contiguous()), _lib.ptr(yb.contiguous()), _lib.ptr(var), b, None, 1.5, 3.0,
                _lib.ptr(sums), _lib.stream_ptr(),
            ),
            "mgp_loss_sums",
        )
        yk2 = yk.reshape(b, R).contiguous()
        _lib.check(
            _lib.fn("column_sums", yk2.dtype)(_lib.ptr(yk2), b, R, _lib.ptr(out[P_YKY0:]), _lib.stream_ptr()),
            "mgp_column_sums",
        )
        out[P_R2_OVER_V] = sums[4]
        out[P_LOG_V] = sums[5]
        out[P_R2] = sums[0]
        out[P_COUNT] = float(b)
    return out, mean, var


def finish_objective(partials: Sequence[float], nn_count: int, loss: str = "lool") -> Dict[str, float]:
    """Global sigma^2 and loss from the (all-reduced) partial sums."""
    p = [float(v) for v in partials]
    n = p[P_COUNT]
    sigma_sq = [yk / (n * nn_count) for yk in p[P_YKY0:]]
    s = sigma_sq[0]
    out = {"sigma_sq": s, "sigma_sq_all": sigma_sq, "count": n, "mse": p[P_R2] / n}
    out["lool"] = p[P_R2_OVER_V] / s + p[P_LOG_V] + n * math.log(s)
    out["objective"] = -out["lool"] if loss == "lool" else -out["mse"]
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
) -> Dict:
    """One LOOCV objective evaluation over all ranks (objective.py:101-103 semantics: the value
    under ``"objective"`` is MINUS the loss).

    ``batch_indices`` / ``nn_indices`` are the global batch unless ``presharded``; each rank
    evaluates its block and the partial sums meet in one all-reduce.  Returns the global scalars
    plus this rank's ``mean`` / ``var`` (which stay sharded, like the reference's results)."""
    if loss not in ("lool", "mse"):
        raise ValueError(f"sharded_loocv supports 'lool' and 'mse', not {loss!r}")
    rank, size = _world(group)
    if not presharded:
        batch_indices = shard_rows(batch_indices, rank, size)
        nn_indices = shard_rows(nn_indices, rank, size)
    partials, mean, var = local_fn(spec, features, targets, batch_indices, nn_indices)
    allreduce_sum_(partials, group)
    out = finish_objective(partials.tolist(), nn_indices.shape[1], loss)
    out.update(mean=mean, var=var, rank=rank, world_size=size)
    return out


def sharded_posterior(spec, test_features, train_features, train_targets, batch_indices, nn_indices, group=None,
                      gather: bool = False):
    """Posterior mean / unscaled variance of this rank's block of the batch; no collective
    unless ``gather`` (then every rank receives the concatenated results, in rank order)."""
    from muygpys_amd.fused import posterior_mean_var

    rank, size = _world(group)
    bi = shard_rows(batch_indices, rank, size)
    ni = shard_rows(nn_indices, rank, size)
    mean, var = posterior_mean_var(spec, test_features, train_features, bi, ni, train_targets)
    if gather and size > 1:
        import torch.distributed as dist

        sizes = chunk_sizes(nn_indices.shape[0], size)
        means = [torch.empty((s,) + tuple(mean.shape[1:]), device=mean.device, dtype=mean.dtype) for s in sizes]
        vars_ = [torch.empty((s,), device=var.device, dtype=var.dtype) for s in sizes]
        dist.all_gather(means, mean.contiguous(), group=group)
        dist.all_gather(vars_, var.contiguous(), group=group)
        mean, var = torch.cat(means), torch.cat(vars_)
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
