"""GPU: the per-rank partial sums of the LOOCV objective (fused kernel + fp64 reductions) and
the combine step reproduce the reference's sigma_sq / lool / mse; shards concatenate to the
serial result (single process; the N>1 collective path is covered over gloo on CPU)."""

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
    assert res["count"] == len(g["batch_idx"])
    # two manual shards + summed partials == the whole
    P = 3
    total = torch.zeros(5, device="cuda", dtype=torch.float64)
    means = []
    for r in range(P):
        p, mean, _ = D.hip_local_partials(spec, X, y, D.shard_rows(bi, r, P), D.shard_rows(ni, r, P))
        total += p
        means.append(mean)
    fin = D.finish_objective(total.tolist(), ni.shape[1])
    assert_close([fin["lool"]], [g["lool"]], rtol, "lool from 3 shards")
    assert_close(torch.cat(means).cpu().numpy(), g["mean"], rtol, "concatenated shard means")


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
