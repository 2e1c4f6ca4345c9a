"""Fused hot-path entry points (one HIP launch from features + indices to mean/var).

These wrap ``mgp_posterior_{f32,f64}`` of the C ABI.  They are what the lazy tensor
handles of the ``hip`` backend call when a MuyGPyS-style caller runs
``make_predict_tensors -> kernel -> posterior_mean / posterior_variance`` (reference
call stack: gp/muygps.py:406-475, gp/kernels/matern.py:148-168, gp/muygps.py:164-259),
and what ``bench.py`` times.
"""

from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import torch

from . import _lib

Number = Union[int, float]


@dataclass
class KernelSpec:
    """Hyper-parameters of one local-GP model, in the reference's vocabulary.

    kernel: "rbf" | "matern05" | "matern15" | "matern25" | "maternInf"
        (_src/gp/kernels/numpy.py:12-31; Matern with fixed nu, gp/kernels/matern.py:61-81), or
        "matern_gen" with ``smoothness`` = nu (numpy.py:34-43: any 0 < nu <= 30 on the fused path, fp32 and -- since
        round 4 -- fp64 tables; :class:`FusedUnsupported` otherwise, and the caller materialises)
    metric: "l2" | "F2"  (gp/deformation/metric.py:237-265)
    length_scale: scalar -> Isotropy (isotropy.py:60-89); sequence of d -> Anisotropy
        (anisotropy.py:43-70)
    noise: scalar -> HomoscedasticNoise; 1-D tensor (n_train,) -> heteroscedastic table
        gathered with nn_indices (tensors/numpy.py:11-15); 2-D tensor (b, k) ->
        already gathered HeteroscedasticNoise tensor (noise/numpy.py:56-67)
    """

    kernel: str = "matern15"
    metric: str = "l2"
    length_scale: Union[Number, Sequence[Number], torch.Tensor] = 1.0
    noise: Union[Number, torch.Tensor] = 0.0
    smoothness: Optional[float] = None

    def kernel_id(self) -> int:
        try:
            return _lib.KERNEL_IDS[self.kernel]
        except KeyError:
            raise ValueError(f"unknown kernel {self.kernel!r}; expected one of {sorted(_lib.KERNEL_IDS)}")

    def metric_id(self) -> int:
        try:
            return _lib.METRIC_IDS[self.metric]
        except KeyError:
            raise ValueError(f"unknown metric {self.metric!r}; expected 'l2' or 'F2'")


class FusedUnsupported(RuntimeError):
    """The fused kernels do not serve this model / shape (general-smoothness Matern beyond nu = 30, or with more
    than 32 slots in a batch too small to compile a kernel for ...): evaluate through the per-function kernels instead."""


# device-resident length scales of host-valued hyper-parameters, keyed by (values, device, dtype): an
# optimiser revisits the same KernelSpec values for mean, variance and scale of one evaluation, and
# a prediction / benchmark loop calls with one spec many times -- no host-to-device copy per call
_LS_CACHE: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
_LS_CACHE_SIZE = 64


def _length_scale_tensor(ls, d: int, like: torch.Tensor) -> torch.Tensor:
    if isinstance(ls, torch.Tensor):
        t = ls.detach().to(device=like.device, dtype=like.dtype).reshape(-1).contiguous()
    else:
        vals = (float(ls),) if isinstance(ls, (int, float)) else tuple(float(v) for v in ls)
        key = (vals, like.device, like.dtype)
        t = _LS_CACHE.get(key)
        if t is None:
            t = torch.tensor(vals, device=like.device, dtype=like.dtype)
            _LS_CACHE[key] = t
            if len(_LS_CACHE) > _LS_CACHE_SIZE:
                _LS_CACHE.popitem(last=False)
        else:
            _LS_CACHE.move_to_end(key)
    if t.numel() != 1 and t.numel() != d:
        raise ValueError(
            f"Difference tensor of shape (..., {d}) must have final dimension size of {t.numel()}"
        )
    return t


class PackedTable:
    """A prepared table (``mgp_table_pack_*``): rows ``[features | responses | pad]`` at a 64-byte
    multiple stride, so that the gather of a neighbour row brings its responses along (two cache
    lines per neighbour instead of three at d = 40, fp32).  Built once per (features, targets) pair
    -- the tables do not change across the objective evaluations of a hyper-parameter search.

    Feature rows that are not whole 16-byte groups (d = 1, 2, 3, 5, 6 ... in fp32; odd d in fp64) are padded with
    zero features up to the next group (``d_kernel``): a zero feature adds nothing to any distance, so the kernels run
    on ``d_kernel`` features (Anisotropy: the padded features get length scale 1) and every shape takes the pipelined
    prepared-table kernels -- the reference's univariate tutorial (d = 1) included."""

    def __init__(self, features: torch.Tensor, targets: Optional[torch.Tensor] = None):
        _lib.require_cuda(features, targets)
        f = (features[:, None] if features.ndim == 1 else features).contiguous()
        t = None
        if targets is not None:
            t = (targets[:, None] if targets.ndim == 1 else targets).to(f.dtype).contiguous()
            if t.shape[0] != f.shape[0]:
                raise ValueError("features and targets differ in row count")
        # the source tensors stay referenced for as long as this table lives: the cache below is keyed on
        # their addresses, which must not be handed to another tensor meanwhile
        self._sources = (features, targets)
        self.n, self.d = f.shape
        group = 16 // f.element_size()
        self.d_kernel = (self.d + group - 1) // group * group
        if self.d_kernel != self.d:
            f = torch.nn.functional.pad(f, (0, self.d_kernel - self.d))
        self.R = 0 if t is None else t.shape[1]
        self.dtype = f.dtype
        self.stride = int(_lib.load().mgp_packed_row_bytes(self.d_kernel, self.R, f.element_size()))
        self.data = torch.empty(self.n * self.stride, dtype=torch.uint8, device=f.device)
        _lib.check(
            _lib.fn("table_pack", f.dtype)(
                _lib.ptr(f), _lib.ptr(t), self.n, self.d_kernel, self.R, _lib.ptr(self.data), self.stride, _lib.stream_ptr()
            ),
            "mgp_table_pack",
        )

    @staticmethod
    def supported(d: int, R: int, k: int, dtype) -> bool:
        es = 4 if dtype == torch.float32 else 8
        dk = (d * es + 15) // 16 * 16 // es  # (rows are padded to whole 16-byte groups)
        if dk > 64:
            return False
        if R * es <= 16 and k + 1 + R <= 64:  # the wave kernels: the responses of a row in one 16-byte slot
            return True
        # fp32 predictions with up to sixteen responses and nn_count <= 64 (BASELINE config 5; round 5): the kernel on the
        # matrix cores' layout reads rows [features | 16 responses] at a 256-byte stride (d = 40)
        return es == 4 and 4 < R <= 16 and k <= 64 and dk >= 8 and (dk + 7) // 8 * 8 != 48

    def kernel_length_scale(self, ls: torch.Tensor) -> torch.Tensor:
        """Per-feature length scales padded to ``d_kernel`` entries (ones: the padded features are all zero)."""
        if ls.numel() == 1 or self.d_kernel == self.d:
            return ls
        return torch.cat([ls, torch.ones(self.d_kernel - self.d, device=ls.device, dtype=ls.dtype)])


# Prepared tables are kept per (features, targets) identity: at most _PACK_CACHE_SIZE entries and
# _PACK_CACHE_BYTES bytes of packed copies (an entry also keeps its source tensors alive -- see
# clear_caches()).  A table packed WITHOUT responses (the query side of a prediction on a separate test
# table: every row is read once per neighbourhood, so its pack is pure overhead kept small, not an
# investment) gets a single slot of its own and never evicts a training table.
_PACK_CACHE: "OrderedDict[tuple, PackedTable]" = OrderedDict()
_PACK_CACHE_SIZE = 4
_PACK_CACHE_BYTES = 8 << 30
_QUERY_PACK: "OrderedDict[tuple, PackedTable]" = OrderedDict()


def _tensor_key(t: Optional[torch.Tensor]):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), t.dtype, t.device)


def pack_table(features: torch.Tensor, targets: Optional[torch.Tensor] = None, query: Optional[bool] = None) -> PackedTable:
    """The prepared table of ``(features, targets)``, cached on the tensors' identity and version
    (an in-place edit of either tensor invalidates the entry).  ``query``: the table serves the query
    side only (default: a table without responses does)."""
    key = (_tensor_key(features), _tensor_key(targets))
    cache = _QUERY_PACK if (targets is None if query is None else query) else _PACK_CACHE
    hit = cache.get(key)
    if hit is None:
        hit = PackedTable(features, targets)
        if cache is _QUERY_PACK:
            cache.clear()
        cache[key] = hit
        while len(cache) > 1 and (len(cache) > _PACK_CACHE_SIZE or sum(t.data.numel() for t in cache.values()) > _PACK_CACHE_BYTES):
            cache.popitem(last=False)
    else:
        cache.move_to_end(key)
    return hit


def clear_caches() -> None:
    """Drop the cached prepared tables (and with them the references to their source tensors) and the
    cached device-resident length scales."""
    _PACK_CACHE.clear()
    _QUERY_PACK.clear()
    _LS_CACHE.clear()


def _check_indices(name: str, idx: Optional[torch.Tensor], n: int) -> None:
    """Reference behaviour for an index outside the table: IndexError from the fancy index
    (_src/gp/tensors/numpy.py:47-69).  Costs a device round trip, so only under
    MUYGPYS_HIP_CHECK_INDICES=1 (debugging a neighbour table built on another data set)."""
    if idx is None or idx.numel() == 0:
        return
    lo, hi = int(idx.amin()), int(idx.amax())
    if lo < 0 or hi >= n:
        raise IndexError(f"{name} holds indices in [{lo}, {hi}] but the table has {n} rows")


def _noise_args(noise, b: int, k: int, like: torch.Tensor):
    if isinstance(noise, torch.Tensor) and noise.ndim >= 1:
        _lib.require_cuda(noise)
        nz = noise.to(dtype=like.dtype).contiguous()
        if nz.ndim == 1:
            if nz.shape[0] != like.shape[0]:
                raise ValueError(
                    f"per-training-point noise table holds {nz.shape[0]} entries for {like.shape[0]} training points"
                )
            return _lib.NOISE_TABLE, 0.0, nz
        if nz.shape != (b, k):
            raise ValueError(f"heteroscedastic noise tensor must have shape {(b, k)}, got {tuple(nz.shape)}")
        return _lib.NOISE_BATCH, 0.0, nz
    return _lib.NOISE_SCALAR, float(noise), None


def posterior_mean_var(
    spec: KernelSpec,
    test_features: torch.Tensor,
    train_features: torch.Tensor,
    batch_indices: Optional[torch.Tensor],
    nn_indices: torch.Tensor,
    train_targets: torch.Tensor,
    want_ykinvy: bool = False,
    out_mean: Optional[torch.Tensor] = None,
    out_var: Optional[torch.Tensor] = None,
    info: Optional[torch.Tensor] = None,
    path: str = "auto",
    packed: Union[str, bool] = "auto",
    gathered: bool = False,
):
    """Posterior mean and *unscaled* variance of every batch element, fused.

    Returns ``(mean, var)`` or ``(mean, var, ykinvy)``: mean is ``(b,)`` for 1-D targets
    and ``(b, R)`` for ``(n, R)`` targets (like _muygps_posterior_mean,
    _src/gp/muygps/numpy.py:17-41); var ``(b,)`` equals ``1 - Kcross K^-1 Kcross``
    (:44-67 with Kout = 1); ``ykinvy (b, R)`` holds ``y_r^T K^-1 y_r`` per neighbourhood
    (the summand of _analytic_scale_optim_unnormalized, scale/numpy.py:9-15).

    ``path``: "auto" (dispatcher), or one kernel family by name -- "generic" / "rhs" (parity tests).
    ``packed``: "auto" reads the tables through prepared copies (:class:`PackedTable`, cached per
    tensor) when the shape allows it and the batch touches the table often enough to pay for the
    one-time pack; True forces, False disables.
    ``gathered``: ``train_targets`` is the already gathered ``(b, k[, R])`` tensor ``targets[nn_indices]``
    (what the reference's ``make_*_tensors`` return) instead of the ``(n[, R])`` table.
    """
    _lib.require_cuda(test_features, train_features, batch_indices, nn_indices, train_targets)
    dtype = train_features.dtype
    if test_features.dtype != dtype or train_targets.dtype != dtype:
        raise TypeError("features and targets must share one float dtype")
    fq = test_features.contiguous()
    fn = train_features.contiguous()
    if fq.ndim == 1:
        fq = fq[:, None]
    if fn.ndim == 1:
        fn = fn[:, None]
    d = fn.shape[1]
    if fq.shape[1] != d:
        raise ValueError("test and train features differ in feature count")
    ni = nn_indices.to(torch.int64).contiguous()
    b, k = ni.shape
    bi = None if batch_indices is None else batch_indices.to(torch.int64).contiguous()
    if bi is not None and bi.shape != (b,):
        raise ValueError("batch_indices must have shape (batch_count,)")
    if bi is None and fq.shape[0] < b:
        raise ValueError(f"{b} neighbourhoods but only {fq.shape[0]} query rows (batch_indices is None)")
    if gathered:
        if tuple(train_targets.shape[:2]) != (b, k):
            raise ValueError(f"gathered responses must have shape ({b}, {k}[, R]), got {tuple(train_targets.shape)}")
        squeeze = train_targets.ndim == 2
        tg = train_targets.reshape(b, k, -1).contiguous()
        R = tg.shape[2]
    else:
        squeeze = train_targets.ndim == 1
        tg = (train_targets[:, None] if squeeze else train_targets).contiguous()
        if tg.shape[0] != fn.shape[0]:
            raise ValueError("train_features and train_targets differ in row count")
        R = tg.shape[1]
    if os.environ.get("MUYGPYS_HIP_CHECK_INDICES") == "1":
        _check_indices("nn_indices", ni, fn.shape[0])
        _check_indices("batch_indices", bi, fq.shape[0])
    ls = _length_scale_tensor(spec.length_scale, d, fn)
    mode, eps, nz = _noise_args(spec.noise, b, k, fn)

    mean = out_mean if out_mean is not None else torch.empty((b, R), device=fn.device, dtype=dtype)
    var = out_var if out_var is not None else torch.empty((b,), device=fn.device, dtype=dtype)
    yk = torch.empty((b, R), device=fn.device, dtype=dtype) if want_ykinvy else None
    if path not in ("auto", "generic", "rhs"):
        raise ValueError(f"unknown kernel path {path!r}")
    if spec.kernel == "matern_gen":
        if path != "auto":
            raise ValueError("the general-smoothness Matern goes through the dispatcher (path='auto')")
        if spec.smoothness is None or not float(spec.smoothness) > 0.0:
            raise ValueError(f"kernel 'matern_gen' needs a positive smoothness, got {spec.smoothness}")
        # one entry point for every table form (mgp_posterior_gen_*): prepared tables when they pay off
        use_packed = (packed is not False and b > 0
                      and PackedTable.supported(d, 0 if gathered else R, k + (R if gathered else 0), dtype))
        if use_packed and packed == "auto":
            key = (_tensor_key(train_features), None if gathered else _tensor_key(train_targets))
            rows = fn.shape[0] + (0 if test_features is train_features else fq.shape[0])
            use_packed = key in _PACK_CACHE or b * (k + 1) >= rows // 4
        pn = pq = None
        if use_packed:
            pn = pack_table(train_features, None if gathered else train_targets, query=False)
            pq = pn if test_features is train_features else pack_table(test_features, None)
        lsk = pn.kernel_length_scale(ls) if use_packed else ls
        rc = _lib.fn("posterior_gen", dtype)(
            None if use_packed else _lib.ptr(fq), None if use_packed else _lib.ptr(fn),
            _lib.ptr(pq.data) if use_packed else None, pq.stride if use_packed else 0,
            _lib.ptr(pn.data) if use_packed else None, pn.stride if use_packed else 0,
            pn.d_kernel if use_packed else d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), R, 1 if gathered else 0, mode, eps,
            _lib.ptr(nz), float(spec.smoothness), spec.metric_id(), _lib.ptr(lsk), lsk.numel(),
            _lib.ptr(mean), _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), _lib.stream_ptr(),
        )
        if rc == -2:
            raise FusedUnsupported(f"general-smoothness Matern: no fused kernel for {dtype}, k={k}, R={R}, d={d}")
        _lib.check(rc, "mgp_posterior_gen")
        mean_out = mean.reshape(b) if squeeze else mean.reshape(b, R)
        if want_ykinvy:
            return mean_out, var, (yk.reshape(b) if squeeze else yk)
        return mean_out, var
    rc = -2
    # gathered responses: the FEATURE rows still come from a prepared table (packed without responses)
    use_packed = (path == "auto" and packed is not False and b > 0
                  and PackedTable.supported(d, 0 if gathered else R, k + (R if gathered else 0), dtype))
    if use_packed and packed == "auto":
        # a pack is one pass over the table; worth it once the batch gathers a comparable number of rows
        key = (_tensor_key(train_features), None if gathered else _tensor_key(train_targets))
        # (a separate test table is packed too -- one more pass over ITS rows -- so it counts as table size)
        rows = fn.shape[0] + (0 if test_features is train_features else fq.shape[0])
        use_packed = key in _PACK_CACHE or b * (k + 1) >= rows // 4
    if use_packed and gathered:
        pn = pack_table(train_features, None, query=False)
        pq = pn if test_features is train_features else pack_table(test_features, None)
        lsk = pn.kernel_length_scale(ls)
        rc = _lib.fn("posterior_packed_gathered", dtype)(
            _lib.ptr(pq.data), pq.stride, _lib.ptr(pn.data), pn.stride, pn.d_kernel, _lib.ptr(bi), _lib.ptr(ni), b, k,
            _lib.ptr(tg), R, mode, eps, _lib.ptr(nz), spec.kernel_id(), spec.metric_id(), _lib.ptr(lsk), lsk.numel(),
            _lib.ptr(mean), _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), _lib.stream_ptr(),
        )
    elif use_packed:
        pn = pack_table(train_features, train_targets)
        pq = pn if test_features is train_features else pack_table(test_features, None)
        lsk = pn.kernel_length_scale(ls)
        rc = _lib.fn("posterior_packed", dtype)(
            _lib.ptr(pq.data), pq.stride, _lib.ptr(pn.data), pn.stride, pn.d_kernel, _lib.ptr(bi), _lib.ptr(ni), b, k, R,
            mode, eps, _lib.ptr(nz), spec.kernel_id(), spec.metric_id(), _lib.ptr(lsk), lsk.numel(),
            _lib.ptr(mean), _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), _lib.stream_ptr(),
        )
    if rc == -2:  # MGP_EUNSUPPORTED on the prepared tables (or not tried): the plain tables
        base = {"auto": "posterior", "generic": "posterior_generic", "rhs": "posterior_rhs"}[path]
        if gathered:
            if path != "auto":
                raise ValueError("gathered responses go through the dispatcher (path='auto')")
            base = "posterior_gathered"
        rc = _lib.fn(base, dtype)(
            _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), R,
            mode, eps, _lib.ptr(nz), spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), ls.numel(),
            _lib.ptr(mean), _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), _lib.stream_ptr(),
        )
    _lib.check(rc, "mgp_posterior")
    mean_out = mean.reshape(b) if squeeze else mean.reshape(b, R)
    if want_ykinvy:
        return mean_out, var, (yk.reshape(b) if squeeze else yk)
    return mean_out, var


def loocv_partials(
    spec: KernelSpec,
    train_features: torch.Tensor,
    train_targets: torch.Tensor,
    batch_indices: torch.Tensor,
    nn_indices: torch.Tensor,
    huber_delta: float = 1.5,
    packed: Union[str, bool] = "auto",
    info: Optional[torch.Tensor] = None,
    return_ykinvy: bool = False,
):
    """One shard's LOOCV evaluation in one library call (``mgp_loocv_*``): the fused launch over
    the training table on both sides, then the six fp64 partial sums
    ``[sum r^2/v, sum log v, sum r^2, b, sum pseudo-Huber, sum y^T K^-1 y]`` the losses
    (_src/optimize/loss/numpy.py:22-61) and the analytic scale (scale/numpy.py:11-18) are built
    from -- no host work between the two.  Returns ``(partials float64 (6,), mean (b,), var (b,))``,
    all on the device.  One response."""
    _lib.require_cuda(train_features, train_targets, batch_indices, nn_indices)
    dtype = train_features.dtype
    if train_targets.dtype != dtype:
        raise TypeError("features and targets must share one float dtype")
    fn = (train_features[:, None] if train_features.ndim == 1 else train_features).contiguous()
    tg = train_targets.reshape(train_targets.shape[0], -1).contiguous()
    if tg.shape[1] != 1:
        raise NotImplementedError("the LOOCV losses are defined for a single response (reference: loss/numpy.py:34-61)")
    if tg.shape[0] != fn.shape[0]:
        raise ValueError("train_features and train_targets differ in row count")
    d = fn.shape[1]
    ni = nn_indices.to(torch.int64).contiguous()
    b, k = ni.shape
    bi = batch_indices.to(torch.int64).contiguous()
    if bi.shape != (b,):
        raise ValueError("batch_indices must have shape (batch_count,)")
    if os.environ.get("MUYGPYS_HIP_CHECK_INDICES") == "1":
        _check_indices("nn_indices", ni, fn.shape[0])
        _check_indices("batch_indices", bi, fn.shape[0])
    ls = _length_scale_tensor(spec.length_scale, d, fn)
    mode, eps, nz = _noise_args(spec.noise, b, k, fn)
    mean = torch.empty((b,), device=fn.device, dtype=dtype)
    var = torch.empty((b,), device=fn.device, dtype=dtype)
    yk = torch.empty((b,), device=fn.device, dtype=dtype)
    partials = torch.empty(6, device=fn.device, dtype=torch.float64)
    scratch = _lib.loocv_scratch(fn.device, b)
    tail = (mode, eps, _lib.ptr(nz), spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), ls.numel(), _lib.ptr(mean),
            _lib.ptr(var), _lib.ptr(yk), _lib.ptr(info), float(huber_delta), _lib.ptr(partials), _lib.ptr(scratch),
            _lib.stream_ptr())
    rc = -2
    use_packed = packed is not False and PackedTable.supported(d, 1, k, dtype) and b > 0
    if use_packed and packed == "auto":
        key = (_tensor_key(train_features), _tensor_key(train_targets))
        use_packed = key in _PACK_CACHE or b * (k + 1) >= fn.shape[0] // 4
    if use_packed:
        pn = pack_table(train_features, train_targets)
        lsk = pn.kernel_length_scale(ls)
        tail_packed = tail[:5] + (_lib.ptr(lsk), lsk.numel()) + tail[7:]  # (`tail` stays the plain-table call's)
        rc = _lib.fn("loocv_packed", dtype)(_lib.ptr(pn.data), pn.stride, pn.d_kernel, _lib.ptr(bi), _lib.ptr(ni), b, k,
                                            *tail_packed)
    if rc == -2:
        rc = _lib.fn("loocv", dtype)(_lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), *tail)
    if rc != 0:
        _lib.loocv_scratch_reset()
    _lib.check(rc, "mgp_loocv")
    if return_ykinvy:
        return partials, mean, var, yk
    return partials, mean, var


class LoocvPlan:
    """A prepared LOOCV objective evaluation: everything that does not change between the evaluations of a
    hyper-parameter search -- prepared table, index tensors, output and scratch buffers, the bound C entry point with its
    argument list -- is set up once; :meth:`launch` then costs one ctypes call (= one kernel launch: the fused kernel
    walks the reduction tree itself) and :meth:`wait` reads the six partial sums from PINNED HOST memory the kernel wrote
    them to, by polling: no stream synchronisation, no device-to-host copy.  What an optimiser's loop pays per evaluation
    drops from ~55 us of host work around a 215 us kernel (config 3's strong-scaling shard, 125 k neighbourhoods) to ~10.

    Reference semantics: one call of the objective ``obj_fn(**hyper)`` of ``make_loo_crossval_fn``
    (optimize/objective.py:20-105) up to the partial sums ``[sum r^2/v, sum log v, sum r^2, b, sum pseudo-Huber,
    sum y^T K^-1 y]`` (``distributed.finish_objective`` turns them into sigma^2 and the loss).

    ``length_scale`` per evaluation: a float (Isotropy) or a sequence of d floats (Anisotropy: copied to the device by
    one asynchronous transfer); ``noise``: a float (HomoscedasticNoise), or fixed per plan as a tensor (heteroscedastic
    table ``(n,)`` / batch ``(b, k)``).  One response.

    The length scale of Isotropy and the result slot are single words of pinned host memory the kernel reads /
    writes while it runs, so ONE evaluation is in flight per plan: :meth:`launch` waits for the previous one if the
    caller did not (``wait()``, or -- device-side partials -- whatever consumed them), and it must be called under the
    stream the plan was made on.  ``info`` (the kernels' count of non-positive pivots) is looked at when a sum comes
    back NaN: ``numpy.linalg.LinAlgError`` as everywhere else (``config.state.check_spd``)."""

    def __init__(self, kernel: str, metric: str, train_features: torch.Tensor, train_targets: torch.Tensor,
                 batch_indices: torch.Tensor, nn_indices: torch.Tensor, anisotropic: bool = False,
                 noise_tensor: Optional[torch.Tensor] = None, huber_delta: float = 1.5, packed: Union[str, bool] = "auto",
                 host_result: bool = True):
        import numpy as np

        _lib.require_cuda(train_features, train_targets, batch_indices, nn_indices, noise_tensor)
        _lib.loocv_tree_selfcheck(train_features.device)  # (once per process: in-kernel walk vs the walk by kernels)
        self.spec = KernelSpec(kernel, metric, 1.0, 0.0)
        dtype = train_features.dtype
        if train_targets.dtype != dtype:
            raise TypeError("features and targets must share one float dtype")
        fn = (train_features[:, None] if train_features.ndim == 1 else train_features).contiguous()
        tg = train_targets.reshape(train_targets.shape[0], -1).contiguous()
        if tg.shape[1] != 1:
            raise NotImplementedError("the LOOCV losses are defined for a single response (reference: loss/numpy.py:34-61)")
        self.d = d = fn.shape[1]
        self.ni = nn_indices.to(torch.int64).contiguous()
        self.b, self.k = b, k = self.ni.shape
        self.bi = batch_indices.to(torch.int64).contiguous()
        if self.bi.shape != (b,):
            raise ValueError("batch_indices must have shape (batch_count,)")
        dev = fn.device
        self.dtype, self.device, self.anisotropic = dtype, dev, bool(anisotropic)
        self._keep = (train_features, train_targets, batch_indices, nn_indices, fn, tg, noise_tensor)
        mode, _, nz = _noise_args(0.0 if noise_tensor is None else noise_tensor, b, k, fn)
        self._nz = nz
        self.mean = torch.empty((b,), device=dev, dtype=dtype)
        self.var = torch.empty((b,), device=dev, dtype=dtype)
        self.ykinvy = torch.empty((b,), device=dev, dtype=dtype)
        self.info = torch.zeros(1, device=dev, dtype=torch.int32)
        self.scratch = torch.zeros(int(_lib.load().mgp_loocv_scratch_bytes()), dtype=torch.uint8, device=dev)
        # the six sums: pinned host memory the kernel writes directly (mapped into the device's address space), or a
        # device tensor (a sharded evaluation all-reduces them on the device first)
        self.host_result = bool(host_result) and b > 0
        if self.host_result:
            self._res_t = torch.zeros(8, dtype=torch.float64).pin_memory()
            self._res = self._res_t.numpy()
            self.partials = self._res_t
        else:
            self.partials = torch.zeros(6, device=dev, dtype=torch.float64)
        use_packed = packed is not False and PackedTable.supported(d, 1, k, dtype) and b > 0
        self._table = pack_table(train_features, train_targets) if use_packed else None
        dk = self._table.d_kernel if use_packed else d
        # length scales: Isotropy -- one value in pinned host memory, read once per workgroup; Anisotropy -- the kernels
        # read them per task: a device buffer refreshed by an asynchronous copy from pinned memory
        self._ls_host = torch.ones(dk if anisotropic else 1, dtype=dtype).pin_memory()
        self._ls_np = self._ls_host.numpy()
        self._ls_dev = torch.ones(dk, device=dev, dtype=dtype) if anisotropic else None
        ls_ptr = _lib.ptr(self._ls_dev if anisotropic else self._ls_host)
        ls_count = dk if anisotropic else 1
        self._eps = _lib.C.c_double(0.0)  # (the one argument that changes per evaluation: set in place)
        tail = [mode, self._eps, _lib.ptr(nz), self.spec.kernel_id(), self.spec.metric_id(), ls_ptr, ls_count, _lib.ptr(self.mean),
                _lib.ptr(self.var), _lib.ptr(self.ykinvy), _lib.ptr(self.info), float(huber_delta), _lib.ptr(self.partials),
                _lib.ptr(self.scratch)]
        if use_packed:
            self._fn = _lib.fn("loocv_packed", dtype)
            head = [_lib.ptr(self._table.data), self._table.stride, dk, _lib.ptr(self.bi), _lib.ptr(self.ni), b, k]
        else:
            self._fn = _lib.fn("loocv", dtype)
            head = [_lib.ptr(fn), d, _lib.ptr(self.bi), _lib.ptr(self.ni), b, k, _lib.ptr(tg)]
        # (every argument converted once: an evaluation is one foreign call on ready-made objects; the stream is the
        # one current when the plan was made -- the plan's scratch must not serve two streams anyway)
        self._stream = _lib.stream_ptr()
        self._raw_stream = _lib.raw_stream()
        self._in_flight = None  # an event behind the last launch whose completion nobody has observed yet
        sig = self._fn.argtypes
        self._args = tuple(a if isinstance(a, _lib.C._SimpleCData) or a is None else t(a) for a, t in zip(head + tail + [self._stream], sig))
        self._np = np
        self._launched = False

    def launch(self, length_scale, noise: float = 0.0) -> None:
        """Put one evaluation on the plan's stream (nothing waits, unless the previous evaluation is still running)."""
        if _lib.raw_stream() != self._raw_stream:
            raise RuntimeError("LoocvPlan.launch(): the current stream is not the one the plan was prepared on")
        if self._in_flight is not None:
            # the kernel in flight reads the length-scale word and writes the result slot this call is about to reset
            if self.host_result:
                self.wait()
            else:
                self._in_flight.synchronize()
            self._in_flight = None
        if self.anisotropic:
            ls = self._np.asarray(length_scale, dtype=self._ls_np.dtype).reshape(-1)
            if ls.size != self.d:
                raise ValueError(f"Difference tensor of shape (..., {self.d}) must have final dimension size of {ls.size}")
            self._ls_np[: self.d] = ls
            self._ls_dev.copy_(self._ls_host, non_blocking=True)
        else:
            self._ls_np[0] = length_scale
        self._eps.value = noise
        if self.host_result:
            self._res[3] = 0.0  # (the kernel writes the count last)
        rc = self._fn(*self._args)
        if rc != 0:
            self.scratch.zero_()
        _lib.check(rc, "mgp_loocv")
        self._launched = True
        if self.host_result:
            self._in_flight = True
        elif self.b > 0:
            self._in_flight = torch.cuda.Event()
            self._in_flight.record()

    def _check_spd(self, sums) -> None:
        if sums[0] != sums[0] or sums[1] != sums[1] or sums[5] != sums[5]:  # NaN: some neighbourhood did not factorise
            from muygpys_amd.config import config

            bad = int(self.info.item())
            self.info.zero_()
            if bad and config.state.check_spd:
                _lib._raise_not_spd(bad, "LOOCV evaluation")

    def wait(self, spin_seconds: float = 2.0):
        """The six partial sums of the evaluation last launched, as host floats (numpy float64 array)."""
        if not self._launched:
            raise RuntimeError("LoocvPlan.wait() before launch()")
        if self.b == 0:
            return self._np.zeros(6)
        if not self.host_result:
            sums = self.partials.cpu().numpy()
            self._in_flight = None
            self._check_spd(sums)
            return sums
        import time

        res, want = self._res, float(self.b)
        deadline = None
        while res[3] != want:
            if deadline is None:
                deadline = time.perf_counter() + spin_seconds
            elif time.perf_counter() > deadline:  # (never in a healthy run: fall back to the stream's own completion)
                torch.cuda.current_stream().synchronize()
                if res[3] != want:
                    raise _lib.HipLibraryError("mgp_loocv: the kernel finished without publishing its sums")
        self._in_flight = None
        sums = res[:6].copy()
        self._check_spd(sums)
        return sums

    def evaluate(self, length_scale, noise: float = 0.0):
        self.launch(length_scale, noise)
        return self.wait()


def loocv_value_and_grad(spec: KernelSpec, train_features: torch.Tensor, train_targets: torch.Tensor,
                         batch_indices: torch.Tensor, nn_indices: torch.Tensor, loss: str = "lool",
                         packed: Union[str, bool] = "auto", reduce_fn=None, scale=("analytic", 1),
                         boundary_scale: Optional[float] = None, sigma_noise: Optional[float] = None):
    """A LOOCV loss and its ANALYTIC gradient with respect to the length scale(s) and a homoscedastic noise: one
    forward evaluation (``mgp_loocv_*``) and one backward launch (``mgp_loocv_backward_*``) instead of the ``p + 1``
    forward evaluations per iteration scipy's finite differences cost the reference's L-BFGS-B driver
    (_src/optimize/chassis/numpy.py:57-81).  The reference gets such gradients from torch autograd over its torch
    backend (torch/muygps_layer.py:129-164); here the chain rule is written out.  With r = mean - y, v the unscaled
    variance, s = sigma^2 and (reference: _src/optimize/loss/numpy.py:22-117)

        lool          sum r^2 / (s v) + log(s v)            d/dm = 2 r / (s v)   d/dv = 1/v - r^2 / (s v^2)   d/ds = sum 1/s - r^2 / (s^2 v)
        mse           sum r^2 / n                           d/dm = 2 r / n
        pseudo_huber  d^2 sum sqrt(1 + (r/d)^2) - 1         d/dm = r / sqrt(1 + (r/d)^2)
        looph         sum 2 d^2 (g - 1) + log(s v),         d/dm = 2 r / (g s v)   d/dv = 1/v - r^2 / (g s v^2)   d/ds = sum 1/s - r^2 / (g s^2 v)
                      g = sqrt(1 + r^2 / (d^2 s v))

    (d = ``boundary_scale``: 1.5 / 3.0 by default, as in the reference), sigma^2 enters through
    ``d s / d (y_i^T K_i^-1 y_i) = (ds / df0) / (n k)`` -- the analytic scale, scale/numpy.py:18-34 -- and the
    backward kernel turns the three cotangents into per-neighbourhood partials of d / d length_scale and d / d noise.
    ``reduce_fn`` (sharded batches): sums a float64 device vector over the ranks in place -- applied to every sum the
    cotangents are formed from and to the gradient.

    ``scale`` names the sigma^2: ``("analytic", iteration_count)`` -- the closed form, followed by
    ``iteration_count - 1`` passes of ``s <- (s + f0 / s) / 2`` on the first value ``f0`` (gp/hyperparameter/scale.py:
    205-217 of the reference; ``ds / df0`` by the same recurrence) -- or ``("fixed", value)``: a constant, nothing
    flows through ``y^T K^-1 y`` (``FixedScale``, the reference's ``noop_scale_opt_fn``).

    ``sigma_noise``: the noise inside sigma^2 when it is NOT ``spec.noise`` -- the reference's objective evaluates mean
    and variance at the TRIAL noise of the optimiser but the analytic scale at the model's STORED one
    (gp/hyperparameter/scale.py:206,214 against gp/noise/homoscedastic.py:112-113).  Then ``y^T K^-1 y`` comes from a
    second forward launch at ``sigma_noise`` and its cotangent goes back through a second backward launch there; the
    returned noise gradient is the trial noise's (sigma^2 does not depend on it).

    Returns ``(value, grad_length_scale (numpy, ls_count), grad_noise (float))`` of the LOSS (the objective the
    drivers maximise is its negative)."""
    import copy
    import math

    import numpy as np

    if loss not in ("lool", "mse", "pseudo_huber", "looph"):
        raise NotImplementedError(f"analytic gradients are written out for lool, mse, pseudo_huber and looph, not {loss!r}")
    if isinstance(spec.noise, torch.Tensor) and spec.noise.ndim >= 1:
        raise NotImplementedError("analytic gradients: homoscedastic noise")
    if spec.kernel == "matern_gen":
        raise NotImplementedError("analytic gradients: closed-form kernels (fixed smoothness)")
    dtype = train_features.dtype
    fn = (train_features[:, None] if train_features.ndim == 1 else train_features).contiguous()
    tg = train_targets.reshape(train_targets.shape[0], -1).contiguous()
    if tg.shape[1] != 1:
        raise NotImplementedError("the LOOCV losses are defined for a single response (reference: loss/numpy.py:34-61)")
    d = fn.shape[1]
    ni = nn_indices.to(torch.int64).contiguous()
    bi = batch_indices.to(torch.int64).contiguous()
    b, k = ni.shape
    needs_scale = loss in ("lool", "looph")
    delta = float(boundary_scale) if boundary_scale is not None else (3.0 if loss == "looph" else 1.5)
    partials, mean, var, yk = loocv_partials(spec, train_features, train_targets, bi, ni, packed=packed, return_ykinvy=True,
                                             huber_delta=delta if loss == "pseudo_huber" else 1.5)
    # (sigma_noise given = the noise is a VARIABLE of the caller's function while sigma^2 holds its own, constant one:
    # the cotangent of y^T K^-1 y must not reach the noise gradient even where the two values coincide)
    split_bwd = needs_scale and scale[0] == "analytic" and sigma_noise is not None
    split = split_bwd and float(sigma_noise) != float(spec.noise)
    spec_s = spec
    if split:  # sigma^2 at the stored noise: y^T K^-1 y of a second evaluation
        spec_s = copy.copy(spec)
        spec_s.noise = float(sigma_noise)
        partials_s = loocv_partials(spec_s, train_features, train_targets, bi, ni, packed=packed)[0]
        partials = torch.cat([partials[:5], partials_s[5:6]])
    if reduce_fn is not None:
        reduce_fn(partials)
    A, B, r2sum, n, ph, cy = (float(v) for v in partials.tolist())
    mode, arg = scale
    if mode == "analytic":
        s = f0 = cy / (n * k)
        ds = 1.0                                  # d s / d f0 through the fixed-point passes
        for _ in range(1, int(arg)):
            s, ds = 0.5 * (s + f0 / s), 0.5 * (ds + 1.0 / s - f0 * ds / (s * s))
    elif mode == "fixed":
        s, ds = float(arg), 0.0
    else:
        raise ValueError(f"scale = {scale!r}: ('analytic', iteration_count) or ('fixed', value)")
    r = mean - tg[:, 0][bi]
    dL_ds = 0.0
    gv = None
    if loss == "lool":
        value = A / s + B + n * math.log(s)
        gm = (2.0 / s) * r / var
        gv = 1.0 / var - (r * r) / (s * var * var)
        dL_ds = n / s - A / (s * s)
    elif loss == "mse":
        value = r2sum / n
        gm = (2.0 / n) * r
    elif loss == "pseudo_huber":
        value = ph
        gm = r / torch.sqrt(1.0 + (r / delta) ** 2)
    else:  # looph: not separable in sigma^2 -- its sums are formed here, in fp64, once sigma^2 is known
        r64, v64 = r.double(), var.double()
        g = torch.sqrt(1.0 + r64 * r64 / (delta * delta * s * v64))
        sums = torch.stack([(2.0 * delta * delta * (g - 1.0) + torch.log(s * v64)).sum(), (r64 * r64 / (g * v64)).sum()])
        if reduce_fn is not None:
            reduce_fn(sums)
        value, rgv = (float(x) for x in sums.tolist())
        gm = (2.0 * r64 / (g * s * v64)).to(dtype)
        gv = (1.0 / v64 - r64 * r64 / (g * s * v64 * v64)).to(dtype)
        dL_ds = n / s - rgv / (s * s)
    zeros = torch.zeros_like(var)
    gv = zeros if gv is None else gv
    gyk_value = dL_ds * ds / (n * k) if needs_scale else 0.0
    ls = _length_scale_tensor(spec.length_scale, d, fn)
    info = torch.zeros(1, device=fn.device, dtype=torch.int32)

    def backward(spec_b, gm_b, gv_b, gyk_b, want_noise):
        g_l = torch.zeros((b, ls.numel()), device=fn.device, dtype=dtype)
        g_n = torch.zeros((b, k), device=fn.device, dtype=dtype) if want_noise else None
        rc = _lib.fn("loocv_backward", dtype)(
            _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(tg), _lib.NOISE_SCALAR, float(spec_b.noise), None,
            spec_b.kernel_id(), spec_b.metric_id(), _lib.ptr(ls), ls.numel(), _lib.ptr(gm_b.contiguous()), _lib.ptr(gv_b.contiguous()),
            _lib.ptr(gyk_b), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info), _lib.stream_ptr(),
        )
        _lib.check(rc, "mgp_loocv_backward")
        return g_l, g_n

    if split_bwd:
        g_l, g_n = backward(spec, gm, gv, zeros, True)
        g_l2, _ = backward(spec_s, zeros, zeros, torch.full_like(var, gyk_value), False)
        g_l = g_l + g_l2
    else:
        g_l, g_n = backward(spec, gm, gv, torch.full_like(var, gyk_value) if gyk_value != 0.0 else zeros, True)
    _lib.raise_if_not_spd(info, "LOOCV gradient")
    grad = torch.cat([_lib.column_sums(g_l), _lib.column_sums(g_n.reshape(-1, 1))])  # fp64, deterministic
    if reduce_fn is not None:
        reduce_fn(grad)
    g = grad.cpu().numpy()
    return value, np.asarray(g[:-1], dtype=np.float64), float(g[-1])


def loocv_tree_sums(mean: torch.Tensor, var: torch.Tensor, ykinvy: torch.Tensor, train_targets: torch.Tensor,
                    batch_indices: Optional[torch.Tensor], huber_delta: float = 1.5, leaves=(0, 0)) -> torch.Tensor:
    """The LOOCV partial sums from finished outputs (``mgp_loocv_tree_*``): the reduction tree ``mgp_loocv_*`` walks
    inside its fused launch, as three small launches -- equal bits for equal ``leaves`` = ``_lib.last_loocv_geometry()``
    of that call (csrc/mgp_loocv_tree.h).  ``mean`` / ``var`` / ``ykinvy`` ``(b,)`` as returned by
    :func:`posterior_mean_var` with ``want_ykinvy``; one response."""
    _lib.require_cuda(mean, var, ykinvy, train_targets, batch_indices)
    dtype = mean.dtype
    b = mean.numel()
    tg = train_targets.reshape(-1).to(dtype).contiguous()
    bi = None if batch_indices is None else batch_indices.to(torch.int64).contiguous()
    out = torch.empty(6, device=mean.device, dtype=torch.float64)
    scratch = torch.empty(int(_lib.load().mgp_loocv_scratch_bytes()), dtype=torch.uint8, device=mean.device)
    rc = _lib.fn("loocv_tree", dtype)(_lib.ptr(mean.contiguous()), _lib.ptr(var.contiguous()), _lib.ptr(ykinvy.contiguous()),
                                      _lib.ptr(tg), tg.element_size(), _lib.ptr(bi), b, float(huber_delta), int(leaves[0]), int(leaves[1]), _lib.ptr(out),
                                      _lib.ptr(scratch), _lib.stream_ptr())
    _lib.check(rc, "mgp_loocv_tree")
    return out


def fast_posterior_mean(
    spec: KernelSpec,
    test_features: torch.Tensor,
    train_features: torch.Tensor,
    test_indices: Optional[torch.Tensor],
    closest_set: torch.Tensor,
    coeffs: torch.Tensor,
    closest_neighbor: torch.Tensor,
    out: Optional[torch.Tensor] = None,
):
    """Prediction from precomputed coefficients, fused (``mgp_fast_posterior_mean_*``).

    ``closest_set (b, k)`` is the self-including neighbourhood of each test point's closest
    training point ``closest_neighbor (b,)`` and ``coeffs (n_train, k[, R])`` the coefficient
    table (reference workflow: examples/fast_posterior_mean.py:373-400; maths
    _src/gp/muygps/numpy.py:70-77).  Returns ``(b,)`` or ``(b, R)``."""
    _lib.require_cuda(test_features, train_features, test_indices, closest_set, coeffs, closest_neighbor)
    dtype = train_features.dtype
    fq = (test_features[:, None] if test_features.ndim == 1 else test_features).contiguous()
    fn = (train_features[:, None] if train_features.ndim == 1 else train_features).contiguous()
    d = fn.shape[1]
    ni = closest_set.to(torch.int64).contiguous()
    b, k = ni.shape
    bi = None if test_indices is None else test_indices.to(torch.int64).contiguous()
    squeeze = coeffs.ndim == 2
    co = (coeffs[:, :, None] if squeeze else coeffs).to(dtype).contiguous()
    if co.shape[1] != k:
        raise ValueError(f"coefficient rows hold {co.shape[1]} entries but the neighbourhoods have {k}")
    R = co.shape[2]
    crow = closest_neighbor.to(torch.int64).contiguous()
    ls = _length_scale_tensor(spec.length_scale, d, fn)
    mean = out if out is not None else torch.empty((b, R), device=fn.device, dtype=dtype)
    rc = _lib.fn("fast_posterior_mean", dtype)(
        _lib.ptr(fq), _lib.ptr(fn), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(co), _lib.ptr(crow), R,
        spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), ls.numel(), _lib.ptr(mean), _lib.stream_ptr(),
    )
    if rc == -2:
        raise FusedUnsupported(f"mgp_fast_posterior_mean serves k + 1 <= 64 slots; got nn_count = {k}")
    _lib.check(rc, "mgp_fast_posterior_mean")
    return mean.reshape(b) if squeeze else mean


def fast_coefficients(spec: KernelSpec, train_features: torch.Tensor, train_targets: torch.Tensor,
                      train_nn_indices: torch.Tensor, chunk: int = 262144, fused: bool = True) -> torch.Tensor:
    """Coefficient table ``C_i = (K_i + eps)^-1 y_i`` over the self-including neighbourhoods
    ``[i, nn[i][:-1]]`` (_fast_nn_update + _muygps_fast_posterior_mean_precompute,
    _src/gp/tensors/numpy.py:97-108, _src/gp/muygps/numpy.py:88-95), computed once per model: one
    fused launch (``mgp_fast_coefficients_*``: one response, k <= 62) or, with ``fused=False`` / for
    other shapes, row chunks through the per-function kernels.  Returns ``(n, k)`` or ``(n, k, R)`` together
    with the updated index table ``(n, k)``."""
    from muygpys_amd._src.gp.kernels import hip as K
    from muygpys_amd._src.gp.muygps import hip as M
    from muygpys_amd._src.gp.noise import hip as N
    from muygpys_amd._src.gp.tensors import hip as T

    _lib.require_cuda(train_features, train_targets, train_nn_indices)
    nn_fast = T._fast_nn_update(train_nn_indices.to(torch.int64))
    n, k = nn_fast.shape
    d = 1 if train_features.ndim == 1 else train_features.shape[1]
    squeeze = train_targets.ndim == 1
    R = 1 if squeeze else train_targets.shape[1]
    if R == 1 and fused:
        # one launch: mgp_fast_coefficients_* (fused gather .. LDL^T .. back-substitution)
        fn = (train_features[:, None] if train_features.ndim == 1 else train_features).contiguous()
        tg = train_targets.reshape(-1).to(fn.dtype).contiguous()
        ls = _length_scale_tensor(spec.length_scale, d, fn)
        mode, eps, nz = _noise_args(spec.noise, n, k, fn)
        out = torch.empty((n, k), device=fn.device, dtype=fn.dtype)
        info = torch.zeros(1, device=fn.device, dtype=torch.int32)
        rc = _lib.fn("fast_coefficients", fn.dtype)(
            _lib.ptr(fn), d, _lib.ptr(nn_fast), n, k, _lib.ptr(tg), mode, eps, _lib.ptr(nz),
            spec.kernel_id(), spec.metric_id(), _lib.ptr(ls), ls.numel(), _lib.ptr(out), _lib.ptr(info),
            _lib.stream_ptr(),
        )
        if rc != -2:  # anything but MGP_EUNSUPPORTED (shape outside the fused kernel)
            _lib.check(rc, "mgp_fast_coefficients")
            _lib.raise_if_not_spd(info, "fast_coefficients")
            return (out if squeeze else out[:, :, None]), nn_fast
    out = torch.empty((n, k, R), device=train_features.device, dtype=train_features.dtype)
    aniso = not isinstance(spec.length_scale, (int, float)) and len(spec.length_scale) > 1
    for s in range(0, n, chunk):
        idx = nn_fast[s:s + chunk].contiguous()
        if aniso:
            ls = _length_scale_tensor(spec.length_scale, d, train_features)
            dist = T._reduce(T._pairwise_tensor(train_features, idx), spec.metric_id(), ls)
            Kin = K._apply(dist, spec.kernel, 1.0)
        else:
            ell = float(spec.length_scale if isinstance(spec.length_scale, (int, float)) else spec.length_scale[0])
            scale = 1.0 / ell if spec.metric == "l2" else 1.0 / ell**2
            Kin = K._apply(T._pairwise_distances(train_features, idx, spec.metric), spec.kernel, scale)
        if isinstance(spec.noise, torch.Tensor) and spec.noise.ndim >= 1:
            Kin = N._heteroscedastic_perturb(Kin, spec.noise[idx] if spec.noise.ndim == 1 else spec.noise[s:s + chunk])
        else:
            Kin = N._homoscedastic_perturb(Kin, float(spec.noise))
        Y = train_targets[idx].reshape(idx.shape[0], k, R)
        out[s:s + chunk] = M._solve(Kin, None, Y, want=("coeffs",))[3]
    return (out.reshape(n, k) if squeeze else out), nn_fast
