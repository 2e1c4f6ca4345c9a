"""Host logic of bench.py that needs no GPU: the strong-scaling shards tile the global batch exactly (reference chunk
rule, src/MuyGPyS/_src/mpi_utils.py:36-41), weak-scaling ranks draw different batches, and the byte / flop formulas
are the ones SURVEY.md sec. 8(d) states."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _cfg(**kw):
    import bench

    return dict(bench.CONFIGS[2], points=5000, batch=1003, **kw)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_strong_scaling_shards_tile_the_global_batch(world):
    import bench
    from muygpys_amd.distributed import chunk_sizes

    cpu = torch.device("cpu")
    whole = bench.build_workload(_cfg(), cpu, 0, False, 1, True)
    parts = [bench.build_workload(_cfg(), cpu, r, False, world, True) for r in range(world)]
    assert [p["b"] for p in parts] == chunk_sizes(1003, world)
    assert torch.equal(torch.cat([p["bi"] for p in parts]), whole["bi"])
    assert torch.equal(torch.cat([p["ni"] for p in parts]), whole["ni"])
    for p in parts:
        assert p["mean"].shape == (p["b"], 1) and p["var"].shape == (p["b"],)
        assert torch.equal(p["X"], whole["X"])  # tables replicated


def test_weak_scaling_ranks_draw_their_own_batches():
    import bench

    cpu = torch.device("cpu")
    a = bench.build_workload(_cfg(), cpu, 0, False, 2, False)
    b = bench.build_workload(_cfg(), cpu, 1, False, 2, False)
    assert a["b"] == b["b"] == 1003 and not torch.equal(a["ni"], b["ni"])


def test_algorithmic_bytes_and_flops_formulas():
    import bench

    assert bench.algorithmic_bytes(30, 40, 1, 4) == 5336  # SURVEY.md sec. 8(d), DESIGN.md sec. 4.1
    assert bench.algorithmic_bytes(30, 40, 1, 4, loocv=True) == 5340
    assert bench.algorithmic_bytes(64, 40, 16, 4) == 65 * 160 + 64 * 64 + 8 * 65 + 17 * 4
    k, d = 30, 40
    pairs = k * (k - 1) / 2 + k
    assert bench.algorithmic_flops(k, d, 1) == pairs * 3 * d + 10 * pairs + k**3 / 3 + 2 * k * k + 4 * k
