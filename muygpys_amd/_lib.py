"""ctypes binding of libmuygpys_hip.so (the C ABI in include/muygpys_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this
module raises.  torch is imported first so that the library's libamdhip64 dependency
resolves to the HIP runtime torch already loaded (one runtime per process; device
pointers and streams are then interchangeable).
"""

from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (must precede the dlopen below)

HERE = os.path.dirname(os.path.abspath(__file__))
# MUYGPYS_HIP_LIB points at an alternative build of the same ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("MUYGPYS_HIP_LIB") or os.path.join(HERE, "lib", "libmuygpys_hip.so")
HEADER = os.path.join(HERE, "..", "include", "muygpys_hip.h")

KERNEL_IDS = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4, "matern_gen": 5}
METRIC_IDS = {"l2": 0, "F2": 1}
NOISE_SCALAR, NOISE_TABLE, NOISE_BATCH = 0, 1, 2

_lib = None


class HipLibraryError(RuntimeError):
    pass


def _status_message(rc: int) -> str:
    if rc == -1:
        return "MGP_EINVAL (null pointer / bad size / unknown enum)"
    if rc == -2:
        return "MGP_EUNSUPPORTED (shape outside what the kernels were built for)"
    if rc <= -1000:
        return f"HIP runtime error {-rc - 1000}"
    return f"status {rc}"


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipLibraryError(f"{what} failed: {_status_message(rc)}")


_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_d = C.c_double

_SIGS = {
    "posterior": [_p, _p, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "crosswise_diffs": [_p, _p, _i, _p, _p, _l, _i, _p, _p],
    "pairwise_diffs": [_p, _i, _p, _l, _i, _p, _p],
    "crosswise_dists": [_p, _p, _i, _p, _p, _l, _i, _i, _p, _p],
    "pairwise_dists": [_p, _i, _p, _l, _i, _i, _p, _p],
    "reduce_diffs": [_p, _l, _i, _p, _i, _p, _p],
    "kernel_apply": [_p, _l, _i, _d, _p, _p],
    "perturb": [_p, _l, _i, _i, _d, _p, _p, _p],
    "matern_gen": [_p, _l, _d, _d, _p, _p],
    "solve": [_p, _p, _p, _l, _i, _i, _d, _p, _p, _p, _p, _p, _p],
    "posterior_generic": [_p, _p, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "posterior_rhs": [_p, _p, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "posterior_gathered": [_p, _p, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "posterior_packed": [_p, _l, _p, _l, _i, _p, _p, _l, _i, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "posterior_packed_gathered": [_p, _l, _p, _l, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p],
    "posterior_gen": [_p, _p, _p, _l, _p, _l, _i, _p, _p, _l, _i, _p, _i, _i, _i, _d, _p, _d, _i, _p, _i, _p, _p, _p, _p, _p],
    "table_pack": [_p, _p, _l, _i, _i, _p, _l, _p],
    "loocv": [_p, _i, _p, _p, _l, _i, _p, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _d, _p, _p, _p],
    "loocv_packed": [_p, _l, _i, _p, _p, _l, _i, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _d, _p, _p, _p],
    "loocv_tree": [_p, _p, _p, _p, _l, _p, _l, _d, _i, _i, _p, _p, _p],
    "loss_sums": [_p, _p, _p, _l, _p, _d, _d, _p, _p, _p],
    "column_sums": [_p, _l, _i, _p, _p, _p],
    "loocv_backward": [_p, _i, _p, _p, _l, _i, _p, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "posterior_backward": [_p, _p, _i, _p, _p, _l, _i, _p, _i, _i, _d, _p, _i, _i, _p, _i,
                           _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "fast_coefficients": [_p, _i, _p, _l, _i, _p, _i, _d, _p, _i, _i, _p, _i, _p, _p, _p],
    "fast_posterior_mean": [_p, _p, _i, _p, _p, _l, _i, _p, _p, _i, _i, _i, _p, _i, _p, _p],
}


def exported_names_from_header():
    """Every function the public header declares (used by the CPU symbol test)."""
    with open(HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(mgp_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen the library (once) and attach argtypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hip backend)"
        )
    lib = C.CDLL(LIB_PATH)
    lib.mgp_version.restype = C.c_char_p
    lib.mgp_max_nn_count.argtypes = [_i, _i]
    lib.mgp_max_nn_count.restype = _i
    lib.mgp_max_nn_count_backward.argtypes = [_i]
    lib.mgp_max_nn_count_backward.restype = _i
    lib.mgp_knn_scan_f32.argtypes = [_p, _p, _l, _i, _p, _p, _p, _l, _i, _l, _p, _p, _p, _p]
    lib.mgp_knn_scan_f32.restype = _i
    lib.mgp_knn_scan_bf16x3.argtypes = [_p, _p, _p, _l, _i, _p, _p, _p, _p, _l, _i, _l, _p, _p, _p, _p]
    lib.mgp_knn_scan_bf16x3.restype = _i
    lib.mgp_knn_scan_bf16x2_d8.argtypes = lib.mgp_knn_scan_bf16x3.argtypes
    lib.mgp_knn_scan_bf16x2_d8.restype = _i
    lib.mgp_topk_rows_f32.argtypes = [_p, _l, _i, _l, _i, _p, _p, _p]
    lib.mgp_topk_rows_f32.restype = _i
    lib.mgp_knn_finish_f32.argtypes = [_p, _p, _i, _p, _l, _i, _p, _p, _p, _p]
    lib.mgp_knn_finish_f32.restype = _i
    lib.mgp_posterior_kernel_name.argtypes = [_i, _i, _i, _i, _i, _i, C.c_char_p, _i]
    lib.mgp_posterior_kernel_name.restype = _i
    lib.mgp_allreduce_partials.argtypes = [_p, _i, _p, _p]
    lib.mgp_allreduce_partials.restype = _i
    lib.mgp_last_kernel_name.argtypes = [C.c_char_p, _i]
    lib.mgp_last_kernel_name.restype = _i
    lib.mgp_reduce_scratch_doubles.restype = _i
    for name in ("mgp_loocv_scratch_bytes", "mgp_loocv_scratch_zero_bytes"):
        getattr(lib, name).argtypes = []
        getattr(lib, name).restype = _l
    lib.mgp_last_launch_geometry.argtypes = [C.POINTER(_l), C.POINTER(_i)]
    lib.mgp_last_launch_geometry.restype = _i
    lib.mgp_last_loocv_geometry.argtypes = [C.POINTER(_i), C.POINTER(_i)]
    lib.mgp_last_loocv_geometry.restype = _i
    lib.mgp_loocv_tree_mode_get.argtypes = []
    lib.mgp_loocv_tree_mode_get.restype = _i
    lib.mgp_loocv_tree_mode_set.argtypes = [_i]
    lib.mgp_loocv_tree_mode_set.restype = _i
    lib.mgp_matern_gen_constants.argtypes = [_d, C.POINTER(C.c_double)]
    lib.mgp_matern_gen_constants.restype = _i
    lib.mgp_jit_prepare.argtypes = [_i, _i, _i, _i, _i, _i]
    lib.mgp_jit_prepare.restype = _i
    lib.mgp_jit_prepare_backward.argtypes = [_i, _i, _i, _i]
    lib.mgp_jit_prepare_backward.restype = _i
    lib.mgp_jit_mode.argtypes = []
    lib.mgp_jit_mode.restype = _i
    lib.mgp_jit_loaded_count.argtypes = []
    lib.mgp_jit_loaded_count.restype = _i
    lib.mgp_packed_row_bytes.argtypes = [_i, _i, _i]
    lib.mgp_packed_row_bytes.restype = _l
    for base, sig in _SIGS.items():
        for suf in ("f32", "f64"):
            fn = getattr(lib, f"mgp_{base}_{suf}")
            fn.argtypes = sig
            fn.restype = _i
    _lib = lib
    return lib


def suffix(dtype) -> str:
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise TypeError(f"hip backend supports float32/float64 tensors, got {dtype}")


def fn(base: str, dtype):
    return getattr(load(), f"mgp_{base}_{suffix(dtype)}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def raw_stream() -> int:
    """The current HIP stream of the current device as an integer (torch.cuda.current_stream().cuda_stream without the
    Stream object: 0.2 instead of 1.5 us -- it sits in front of every launch of an optimiser's loop)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:  # (a torch without the private accessor)
        return torch.cuda.current_stream().cuda_stream


def stream_ptr():
    return C.c_void_p(raw_stream())


def served_by(d: int, k: int, R: int, dtype, packed: bool = False, path: str = "auto") -> str:
    """Name of the kernel instantiation the dispatcher picks for a shape (``mgp_posterior_kernel_name``)."""
    buf = C.create_string_buffer(256)
    es = 4 if dtype == torch.float32 else 8
    rc = load().mgp_posterior_kernel_name(es, d, k, R, int(bool(packed)), {"auto": 0, "generic": 1, "rhs": 2}[path], buf, 256)
    if rc == -2 and packed:
        return served_by(d, k, R, dtype, False, path)
    check(rc, "mgp_posterior_kernel_name")
    return buf.value.decode()


def last_kernel() -> str:
    """The instantiation this thread's most recent fused call actually launched (``mgp_last_kernel_name``)."""
    buf = C.create_string_buffer(256)
    check(load().mgp_last_kernel_name(buf, 256), "mgp_last_kernel_name")
    return buf.value.decode()


def reduce_scratch(device):
    """Per-call scratch of the deterministic two-stage reductions (mgp_loss_sums_* / mgp_column_sums_*);
    it comes from torch's stream-aware caching allocator, so concurrent streams never share one."""
    return torch.empty(load().mgp_reduce_scratch_doubles(), dtype=torch.float64, device=device)


# scratch of the one-launch LOOCV evaluation (mgp_loocv_*): its counters must be zero when a call starts and every call
# leaves them zero, so a buffer is zeroed ONCE and then reused by every later evaluation on the same
# stream (the inner loop of a hyper-parameter search: no allocation, no memset, one launch per evaluation).
_LOOCV_SCRATCH: "dict" = {}


def loocv_scratch(device, b: int = 0) -> "torch.Tensor":
    """The zero-initialised, per-(device, stream) scratch of ``mgp_loocv_*`` (uint8, 256-byte aligned by torch's
    allocator).  Streams never share one (two evaluations in flight would share counters)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
    buf = _LOOCV_SCRATCH.get(key)
    if buf is None:
        if len(_LOOCV_SCRATCH) >= 16:  # (a search evaluates one batch size; keep the cache from growing without bound)
            _LOOCV_SCRATCH.pop(next(iter(_LOOCV_SCRATCH)))
        buf = torch.zeros(int(load().mgp_loocv_scratch_bytes()), dtype=torch.uint8, device=dev)
        _LOOCV_SCRATCH[key] = buf
    return buf


def last_loocv_geometry():
    """(grid, nh): the leaves of the reduction tree the last ``mgp_loocv_*`` call of this thread walked inside its fused
    launch; (0, 0): the three-launch walk on the canonical leaves served it."""
    g, n = C.c_int(0), C.c_int(0)
    check(load().mgp_last_loocv_geometry(C.byref(g), C.byref(n)), "mgp_last_loocv_geometry")
    return g.value, n.value


TREE_MODES = ("tickets", "fenced", "three_launch")
_TREE_CHECKED = {"done": False, "fell_back": False}


def loocv_tree_mode() -> str:
    return TREE_MODES[load().mgp_loocv_tree_mode_get()]


def set_loocv_tree_mode(mode: str) -> None:
    """``tickets`` | ``fenced`` | ``three_launch`` (include/muygpys_hip.h: mgp_loocv_tree_mode_set); process-wide."""
    check(load().mgp_loocv_tree_mode_set(TREE_MODES.index(mode)), "mgp_loocv_tree_mode_set")


def loocv_tree_selfcheck(device=None, rounds: int = 6, force: bool = False) -> bool:
    """Once per process, before the first prepared LOOCV evaluation: the in-kernel walk of the reduction tree (whose
    hand-off between workgroups relies on write-through stores and a relaxed ticket, csrc/mgp_loocv_tree.h) against the
    walk by three kernels over the same leaves, on a small synthetic shard, ``rounds`` evaluations at three grid
    sizes -- the sums must agree BIT FOR BIT.  If they ever differ (a driver / compiler that reorders what this GPU
    model and ROCm release do not), the process switches to ``three_launch`` for good and says so; nothing else
    changes for the caller (same sums, two small launches more per evaluation).  Skipped when the form was chosen
    explicitly (MUYGPYS_HIP_LOOCV_TREE) or is not the in-kernel default.  Returns True when the in-kernel walk stays."""
    import os
    import warnings

    if _TREE_CHECKED["done"] and not force:
        return not _TREE_CHECKED["fell_back"]
    _TREE_CHECKED["done"] = True
    if (os.environ.get("MUYGPYS_HIP_LOOCV_TREE") and not force) or loocv_tree_mode() == "three_launch":
        return loocv_tree_mode() != "three_launch"
    import numpy as np

    from muygpys_amd import fused

    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    mode = loocv_tree_mode()
    rng = np.random.default_rng(20251004)
    n, k, d = 6000, 30, 40
    X = torch.from_numpy(rng.standard_normal((n, d), dtype=np.float32)).to(dev)
    y = torch.from_numpy(rng.standard_normal(n, dtype=np.float32)).to(dev)
    ok = True
    for b in (64, 4097, 30000):  # one level-2 block / a ragged grid / the full persistent grid
        bi = torch.from_numpy(rng.integers(0, n, size=b)).to(dev)
        ni = torch.from_numpy(rng.integers(0, n, size=(b, k))).to(dev)
        spec = fused.KernelSpec("matern15", "l2", 5.0, 1e-2)
        try:
            set_loocv_tree_mode("three_launch")
            ref = fused.loocv_partials(spec, X, y, bi, ni, packed=False)[0].cpu().numpy()
        finally:
            set_loocv_tree_mode(mode)
        for _ in range(rounds):
            got = fused.loocv_partials(spec, X, y, bi, ni, packed=False)[0].cpu().numpy()
            ok = ok and got.tobytes() == ref.tobytes()
    if not ok:
        _TREE_CHECKED["fell_back"] = True
        set_loocv_tree_mode("three_launch")
        warnings.warn("muygpys_amd: the one-launch LOOCV reduction tree disagreed with its three-launch walk on this "
                      "device / driver; using MUYGPYS_HIP_LOOCV_TREE=three_launch for the rest of the process",
                      RuntimeWarning, stacklevel=2)
    return ok


def last_launch_geometry():
    """(workgroups, dynamic LDS bytes per workgroup) of this thread's last wave-kernel launch."""
    g, n = C.c_int64(0), C.c_int(0)
    check(load().mgp_last_launch_geometry(C.byref(g), C.byref(n)), "mgp_last_launch_geometry")
    return g.value, n.value


def loocv_scratch_reset() -> None:
    """Forget every cached LOOCV scratch (after a failed launch their counters may be dirty)."""
    _LOOCV_SCRATCH.clear()


def column_sums(x2: "torch.Tensor") -> "torch.Tensor":
    """Deterministic fp64 column sums of a contiguous (rows, cols) device tensor (``mgp_column_sums_*``)."""
    rows, cols = x2.shape
    out = torch.zeros(cols, device=x2.device, dtype=torch.float64)
    if rows and cols:
        scratch = reduce_scratch(x2.device)
        check(
            fn("column_sums", x2.dtype)(ptr(x2), rows, cols, ptr(out), ptr(scratch), stream_ptr()),
            "mgp_column_sums",
        )
    return out


def loss_sums(pred, target, var, scale_dev, huber_delta: float, looph_delta: float) -> "torch.Tensor":
    """The six fp64 loss sums of ``mgp_loss_sums_*`` (deterministic two-stage reduction)."""
    out = torch.empty(6, device=pred.device, dtype=torch.float64)
    scratch = reduce_scratch(pred.device)
    check(
        fn("loss_sums", pred.dtype)(
            ptr(pred), ptr(target), ptr(var), pred.numel(), ptr(scale_dev), float(huber_delta), float(looph_delta),
            ptr(out), ptr(scratch), stream_ptr(),
        ),
        "mgp_loss_sums",
    )
    return out


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise TypeError("hip backend functions take torch tensors on a ROCm device ('cuda')")


_SPD_PENDING = []   # deferred mode: [(pinned int32 tensor, event, what)] of the launch not yet looked at
_SPD_SLOTS = []     # two pinned slots, used in turn


def _raise_not_spd(bad: int, what: str) -> None:
    import numpy as np

    raise np.linalg.LinAlgError(
        f"{what}: {bad} neighbourhood(s) are not positive definite (singular or indefinite K + noise)"
    )


def flush_spd_checks() -> None:
    """Deferred mode (config.state.check_spd == "deferred"): look at the counter of the last checked launch now."""
    while _SPD_PENDING:
        host, event, what = _SPD_PENDING.pop(0)
        event.synchronize()
        bad = int(host[0])
        if bad:
            _raise_not_spd(bad, what + " (reported one call late: deferred check)")


def raise_if_not_spd(info, what: str) -> None:
    """Reference behaviour for an unsolvable local system: numpy.linalg.LinAlgError from
    ``linalg.solve`` (_src/gp/muygps/numpy.py:37).  ``info`` is the device counter the kernels
    increment per neighbourhood with a non-positive Cholesky pivot."""
    from muygpys_amd.config import config

    mode = config.state.check_spd
    if info is None or not mode:
        return
    if mode == "deferred":
        import torch

        flush_spd_checks()  # the previous launch's counter: its copy is behind us on the stream
        if len(_SPD_SLOTS) < 2:
            _SPD_SLOTS.append(torch.zeros(1, dtype=torch.int32).pin_memory())
        host = _SPD_SLOTS[0]
        _SPD_SLOTS.reverse()
        host.copy_(info, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        _SPD_PENDING.append((host, event, what))
        return
    bad = int(info.item())
    if bad:
        _raise_not_spd(bad, what)
