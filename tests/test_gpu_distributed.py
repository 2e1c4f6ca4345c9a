"""GPU: the per-rank partial sums of the LOOCV objective (fused kernel + fp64 reductions) and
the combine step reproduce the reference's sigma_sq / lool / mse; shards concatenate to the
serial result (single process; the N>1 collective path is covered over gloo on CPU)."""

import os

import numpy as np
import pytest

from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_partials_and_finish(golden, dtype):
    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec

    g, meta = golden, golden["meta"]
    if "lool" not in g:
        pytest.skip("single-response losses only")
    if dtype == "float32" and meta["d"] < 10 and meta["noise"] < 1e-4 and not meta.get("hetero"):
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too)")
    td = getattr(torch, dtype)
    noise = to_dev(g["noise_table"], td) if meta.get("hetero") else meta["noise"]
    spec = KernelSpec(meta["kernel"], meta["metric"], meta["length_scale"], noise)
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    res = D.sharded_loocv(spec, X, y, bi, ni)
    rtol = RTOL[dtype]
    assert_close([res["sigma_sq"]], g["sigma_sq"], rtol, "sigma_sq")
    assert_close([res["lool"]], [g["lool"]], rtol, "lool")
    assert_close([res["mse"]], [g["mse"]], rtol, "mse")
    assert_close([res["pseudo_huber"]], [g["huber"]], rtol, "pseudo_huber")
    assert_close([D.sharded_loocv(spec, X, y, bi, ni, loss="looph")["looph"]], [g["looph"]], rtol, "looph")
    assert res["count"] == len(g["batch_idx"])
    # two manual shards + summed partials == the whole
    P = 3
    total = torch.zeros(D.P_YKY0 + 1, device="cuda", dtype=torch.float64)
    means = []
    for r in range(P):
        p, mean, _ = D.hip_local_partials(spec, X, y, D.shard_rows(bi, r, P), D.shard_rows(ni, r, P))
        total += p
        means.append(mean)
    fin = D.finish_objective(total.tolist(), ni.shape[1])
    assert_close([fin["lool"]], [g["lool"]], rtol, "lool from 3 shards")
    assert_close(torch.cat(means).cpu().numpy(), g["mean"], rtol, "concatenated shard means")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("packed", [False, True])
def test_loocv_call_equals_the_three_call_composition(dtype, packed):
    """``mgp_loocv_*`` (one call, since round 5 one LAUNCH: the fused kernel walks the reduction tree itself) against
    ``mgp_posterior_*`` + ``mgp_loocv_tree_*`` (the same tree, the same leaves, walked by three small launches over the
    finished outputs):
    the six sums agree to the last bit; and against the other fixed-order reductions of the library
    (``mgp_loss_sums_*`` / ``mgp_column_sums_*``: another summation order) to fp64 rounding."""
    from muygpys_amd import _lib
    from muygpys_amd.fused import KernelSpec, loocv_partials, loocv_tree_sums, posterior_mean_var

    td = getattr(torch, dtype)
    g = torch.Generator(device="cuda").manual_seed(11)
    n, d, k, b = 30000, 40, 30, 20011
    X = torch.randn((n, d), device="cuda", dtype=td, generator=g)
    y = torch.sin(X[:, 0]) + 0.1 * torch.randn((n,), device="cuda", dtype=td, generator=g)
    bi = torch.randperm(n, device="cuda", generator=g)[:b]
    ni = torch.randint(0, n - 1, (b, k), device="cuda", generator=g)
    ni = ni + (ni >= bi[:, None])
    spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
    p, mean, var = loocv_partials(spec, X, y, bi, ni, huber_delta=1.5, packed=packed)
    leaves = _lib.last_loocv_geometry()  # (the workgroups of the fused launch: the leaves of its tree)
    m2, v2, yk2 = posterior_mean_var(spec, X, X, bi, ni, y, want_ykinvy=True, packed=packed)
    assert torch.equal(mean, m2) and torch.equal(var, v2)
    expect = loocv_tree_sums(m2, v2, yk2, y, bi, 1.5, leaves=leaves)
    assert torch.equal(p, expect), (p.tolist(), expect.tolist())
    sums = _lib.loss_sums(m2.contiguous(), y[bi].contiguous(), v2, None, 1.5, 3.0)
    yks = _lib.column_sums(yk2.reshape(b, 1).contiguous())
    other = torch.stack([sums[4], sums[5], sums[0], torch.tensor(float(b), device="cuda", dtype=torch.float64),
                         sums[2], yks[0]])
    torch.testing.assert_close(p, other, rtol=1e-12, atol=1e-9)
    # and against plain fp64 torch arithmetic on the same outputs
    r = m2.double() - y[bi].double()
    ref = torch.stack([(r * r / v2.double()).sum(), v2.double().log().sum(), (r * r).sum()])
    torch.testing.assert_close(p[:3], ref, rtol=1e-12, atol=0)
    # an empty shard contributes zeros
    p0, m0, _ = loocv_partials(spec, X, y, bi[:0], ni[:0], packed=packed)
    assert m0.shape == (0,) and torch.equal(p0, torch.zeros(6, device="cuda", dtype=torch.float64))


def test_sharded_batch_nns_concatenates_to_the_unsharded_search():
    """Three emulated ranks (reference chunk rule, remainder to the last ranks) each search their
    block of the batch; concatenated in rank order the result is the single-process search."""
    from muygpys_amd.distributed import chunk_sizes, sharded_batch_nns
    from muygpys_amd.neighbors import NN_Wrapper

    g = torch.Generator().manual_seed(5)
    X = torch.randn(20000, 16, generator=g).cuda()
    nbrs = NN_Wrapper(X, 12)
    bi = torch.randperm(20000, generator=g)[:1001].cuda()
    ref_i, ref_d = nbrs.get_batch_nns(bi)
    parts = [sharded_batch_nns(nbrs, bi, rank=r, world_size=3) for r in range(3)]
    assert [p[0].shape[0] for p in parts] == chunk_sizes(1001, 3)
    assert torch.equal(torch.cat([p[0] for p in parts]), bi)
    assert torch.equal(torch.cat([p[1] for p in parts]), ref_i)
    torch.testing.assert_close(torch.cat([p[2] for p in parts]), ref_d)


def _gpu_rank(rank, world, port, q):
    """One of two processes sharing cuda:0 (a 1-GPU box): HIP partial sums + a real collective."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from muygpys_amd import distributed as D
        from muygpys_amd.fused import KernelSpec
        from muygpys_amd.gp import MuyGPS
        from muygpys_amd.gp.deformation import Isotropy, l2
        from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
        from muygpys_amd.gp.kernels import Matern
        from muygpys_amd.gp.noise import HomoscedasticNoise
        from tests.conftest import load_golden

        g = load_golden("m15_iso_knn_k30_d40_c2")
        td = torch.float64
        X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
        bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
        spec = KernelSpec(g["meta"]["kernel"], g["meta"]["metric"], g["meta"]["length_scale"], g["meta"]["noise"])
        res = D.sharded_loocv(spec, X, y, bi, ni, loss="looph")
        model = MuyGPS(Matern(smoothness=Parameter(1.5), deformation=Isotropy(l2, length_scale=Parameter(3.0, (0.5, 20.0)))),
                       noise=HomoscedasticNoise(g["meta"]["noise"]), scale=AnalyticScale())
        opt = D.optimize_sharded(model, X, y, bi, ni, optimizer="lbfgsb")
        q.put((rank, res["lool"], res["sigma_sq"], res["looph"], float(opt.kernel.deformation.length_scale())))
    finally:
        dist.destroy_process_group()


def test_two_processes_hip_partials_and_collective_match_serial():
    """HIP partial sums and a real all-reduce together (two ranks on this box's GPU), the functor
    layer's L-BFGS-B under sharded reductions included: the same numbers as one process."""
    import socket

    import torch.multiprocessing as mp

    from muygpys_amd import distributed as D
    from muygpys_amd.fused import KernelSpec
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.optimize import L_BFGS_B_optimize
    from tests.conftest import load_golden

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = load_golden("m15_iso_knn_k30_d40_c2")
    assert got[0][1:] == got[1][1:], "every rank must hold the same global scalars"
    np.testing.assert_allclose(got[0][1], g["lool"], rtol=1e-8)
    np.testing.assert_allclose(got[0][2], g["sigma_sq"][0], rtol=1e-8)
    np.testing.assert_allclose(got[0][3], g["looph"], rtol=1e-8)
    td = torch.float64
    X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
    bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
    model = MuyGPS(Matern(smoothness=Parameter(1.5), deformation=Isotropy(l2, length_scale=Parameter(3.0, (0.5, 20.0)))),
                   noise=HomoscedasticNoise(g["meta"]["noise"]), scale=AnalyticScale())
    cw, pw, bt, bnt = model.make_train_tensors(bi, ni, X, y)
    serial = L_BFGS_B_optimize(model, bt, bnt, cw, pw)
    np.testing.assert_allclose(got[0][4], float(serial.kernel.deformation.length_scale()), rtol=1e-6)


def test_mgp_allreduce_partials_over_a_real_rccl_communicator():
    """The C-ABI collective (mgp_allreduce_partials: RCCL opened with dlopen, communicator handed over as void*):
    a one-rank communicator made through RCCL's own API sums a partial vector in place (a single-GPU box cannot
    hold two ranks -- RCCL refuses two ranks on one device -- so what this pins is the binding: library lookup,
    argument order, data type and reduction codes, stream)."""
    import ctypes as C

    from muygpys_amd import _lib

    lib = _lib.load()
    rccl = None
    for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("no RCCL library on this machine")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        x = torch.tensor([1.5, -2.0, 3.25, 7.0, 0.0, 1e300], device="cuda", dtype=torch.float64)
        want = x.clone()
        rc = lib.mgp_allreduce_partials(_lib.ptr(x), x.numel(), comm, _lib.stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0, rc
        assert torch.equal(x, want)
        assert lib.mgp_allreduce_partials(None, 6, comm, _lib.stream_ptr()) == -1
        assert lib.mgp_allreduce_partials(_lib.ptr(x), 0, comm, _lib.stream_ptr()) == -1
        assert lib.mgp_allreduce_partials(_lib.ptr(x), 6, None, _lib.stream_ptr()) == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def _nccl_one_rank(port, q):
    """torch.distributed over RCCL (`nccl`) with ONE rank on cuda:0, every collective of the path forced on
    (MUYGPYS_HIP_FORCE_COLLECTIVES): what the 8-GPU scaling run executes, minus the other seven GPUs."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MUYGPYS_HIP_FORCE_COLLECTIVES="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import bench
        from muygpys_amd import distributed as D
        from muygpys_amd.fused import KernelSpec
        from muygpys_amd.gp import MuyGPS
        from muygpys_amd.gp.deformation import Isotropy, l2
        from muygpys_amd.gp.hyperparameter import AnalyticScale, Parameter
        from muygpys_amd.gp.kernels import Matern
        from muygpys_amd.gp.noise import HomoscedasticNoise
        from tests.conftest import load_golden

        assert D._collectives_on() and dist.get_backend() == "nccl"
        g = load_golden("m15_iso_knn_k30_d40_c2")
        td = torch.float64
        X, y = to_dev(g["features"], td), to_dev(g["targets"], td)
        bi, ni = to_dev(g["batch_idx"]), to_dev(g["nn_idx"])
        spec = KernelSpec(g["meta"]["kernel"], g["meta"]["metric"], g["meta"]["length_scale"], g["meta"]["noise"])
        out = {}
        # the six sums all-reduced on the device through RCCL, then the second (looph) all-reduce
        res = D.sharded_loocv(spec, X, y, bi, ni, loss="looph")
        out["lool"], out["sigma_sq"], out["looph"] = res["lool"], res["sigma_sq"], res["looph"]
        # broadcasts (sampled hyper-parameters / start points / the Bayes driver's seed): device tensors under nccl
        out["bcast"] = D.broadcast_scalar(3.25)
        out["bvec"] = D.broadcast_vector([1.0, 2.5, -4.0]).tolist()
        out["seed"] = D.synchronized_seed()
        # the functor layer's L-BFGS-B under sharded reductions (every loss / scale sum all-reduced over RCCL)
        model = MuyGPS(Matern(smoothness=Parameter(1.5), deformation=Isotropy(l2, length_scale=Parameter(3.0, (0.5, 20.0)))),
                       noise=HomoscedasticNoise(g["meta"]["noise"]), scale=AnalyticScale())
        opt = D.optimize_sharded(model, X, y, bi, ni, optimizer="lbfgsb")
        out["ls"] = float(opt.kernel.deformation.length_scale())
        # prediction with the padded equal-size gather
        mean, var = D.sharded_posterior(spec, X, X, y, bi, ni, gather=True)
        out["mean"], out["var"] = mean.cpu().numpy(), var.cpu().numpy()
        # bench.py's own collectives (barrier, MAX all-reduce of the elapsed time, the ranks-seen all-reduce)
        elapsed, kern_ms, ranks_seen, _ = bench.time_steps(lambda: D.sharded_loocv(spec, X, y, bi, ni, presharded=True), 2, 3, dist,
                                                           "nccl", torch.device("cuda", 0))
        out["ranks_seen"], out["steps"] = ranks_seen, len(kern_ms)
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_every_collective_of_the_path_through_torchs_rccl_with_one_rank():
    """RCCL has only ever run where there is more than one GPU -- the driver's scaling bench.  A world-size-1 ``nccl``
    process group on this box's GPU executes the same calls on device tensors: init_process_group(device_id=...), the
    all-reduce of the partial sums, the broadcasts, L-BFGS-B under sharded reductions, the padded all_gather_into_tensor
    of sharded_posterior(gather=True), and bench.py's barrier / MAX / ranks-seen collectives.  Values: the fixture's."""
    import socket

    import torch.multiprocessing as mp

    from tests.conftest import load_golden

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank, args=(port, q))
    p.start()
    out = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    g = load_golden("m15_iso_knn_k30_d40_c2")
    np.testing.assert_allclose(out["lool"], g["lool"], rtol=1e-8)
    np.testing.assert_allclose(out["sigma_sq"], g["sigma_sq"][0], rtol=1e-8)
    np.testing.assert_allclose(out["looph"], g["looph"], rtol=1e-8)
    assert out["bcast"] == 3.25 and out["bvec"] == [1.0, 2.5, -4.0] and 0 <= out["seed"] < 2**31
    assert 0.5 <= out["ls"] <= 20.0
    np.testing.assert_allclose(out["mean"], g["mean"], rtol=1e-5, atol=1e-5 * np.sqrt(np.mean(g["mean"] ** 2)))
    np.testing.assert_allclose(out["var"], g["var_unscaled"], rtol=1e-5)
    assert out["ranks_seen"] == 1 and out["steps"] == 3
