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


_RANK_SCRIPT = """
import os, sys, time
import torch.distributed as dist
rank = int(os.environ["RANK"])
if rank == 1 and os.environ.get("FAIL_BEFORE_COLLECTIVE") == "1":
    raise RuntimeError("rank 1 dies before its first collective")
dist.init_process_group("gloo")
dist.barrier()          # rank 0 blocks here for ever when rank 1 never arrives
dist.destroy_process_group()
"""


def _spawn_ranks(tmp_path, n, fail):
    import os
    import subprocess
    import sys

    import bench

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    port = bench._free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FAIL_BEFORE_COLLECTIVE="1" if fail else "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stderr=subprocess.DEVNULL))
    return procs


def test_a_rank_that_dies_before_its_first_collective_takes_the_run_down_quickly(tmp_path):
    """Round-5 review: the parent waited for the ranks in order, so rank 0 sat in its collective until the watchdog.
    Now: non-zero within seconds, no sibling left behind (world size 2 over gloo; rank 1 raises before
    init_process_group, rank 0 is blocked in the rendezvous / barrier)."""
    import time

    import bench

    procs = _spawn_ranks(tmp_path, 2, fail=True)
    t0 = time.monotonic()
    rc = bench.supervise_ranks(procs, timeout_s=120.0)
    assert rc != 0
    assert time.monotonic() - t0 < 30.0
    assert all(p.poll() is not None for p in procs)


def test_supervise_ranks_success_and_timeout(tmp_path):
    import subprocess
    import sys
    import time

    import bench

    assert bench.supervise_ranks(_spawn_ranks(tmp_path, 2, fail=False), timeout_s=120.0) == 0
    sleeper = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"]) for _ in range(2)]
    t0 = time.monotonic()
    assert bench.supervise_ranks(sleeper, timeout_s=1.0, grace_s=2.0) == 124
    assert time.monotonic() - t0 < 20.0 and all(p.poll() is not None for p in sleeper)
