"""bench.py under test: (1) every kernel instantiation a bench line times is one the oracle has checked, through the
exact call bench.py makes; (2) the N > 1 branch (launch_ranks, torch.distributed.run, weak and strong scaling) runs
here, on one device over gloo, before it runs on the 8-GPU node."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.util import RTOL, assert_close

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _oracle_checked_kernel(cid: int, route: str = "fused", **override) -> str:
    """One step of bench.py's own callable for a BASELINE config on a reduced table / batch (the instantiation the
    dispatcher picks for these configs depends on shape, dtype and table form only), compared with the oracle;
    returns the name of the instantiation that was launched."""
    import bench
    from muygpys_amd import _lib
    from muygpys_amd.fused import PackedTable, clear_caches, pack_table

    cfg = dict(bench.CONFIGS[cid], points=30_000, batch=6_000)
    cfg.update(override)
    dev = torch.device("cuda", 0)
    w = bench.build_workload(cfg, dev, 0, False)
    use_packed = PackedTable.supported(w["d"], w["R"], w["k"], w["td"])
    if use_packed and route == "fused":
        pack_table(w["X"], w["y"])
    from muygpys_amd.config import config

    lazy_before = config.state.lazy_tensors
    try:
        step = bench.make_step(cfg, w, route, "auto", use_packed if route == "fused" else "auto")  # (dropin: switches lazy handles on)
        last = step()
        torch.cuda.synchronize()
    finally:
        config.state.lazy_tensors = lazy_before
    name = _lib.last_kernel()
    mean, var = bench._outputs_of(cfg, w, route, last)
    X = w["X"].double().cpu().numpy()
    y = w["y"].double().cpu().numpy()
    bi, ni = w["bi"].cpu().numpy(), w["ni"].cpu().numpy()
    ls = np.asarray(w["ls"], dtype=np.float64) if isinstance(w["ls"], list) else float(w["ls"])
    ospec = orc.Spec(cfg["kernel"], cfg["metric"], ls, cfg["noise"])
    pick = np.unique(np.linspace(0, w["b"] - 1, 600).round().astype(int))
    m_ref, v_ref = orc.posterior_mean_var(ospec, X, X, bi[pick], ni[pick], y)
    rtol = RTOL["float32" if cfg["dtype"] == "f32" else "float64"]
    assert_close(mean.double().cpu().numpy().reshape(w["b"], -1)[pick], m_ref.reshape(len(pick), -1), rtol, f"mean [{name}]")
    assert_close(var.double().cpu().numpy()[pick], v_ref, rtol, f"var [{name}]")
    chk = bench.spot_check(cfg, w, mean, var)
    assert chk["ok"], chk
    clear_caches()
    return name


@pytest.mark.parametrize("cid", [2, 3, 4, 5])
def test_bench_step_of_every_config_matches_the_oracle(cid):
    name = _oracle_checked_kernel(cid)
    assert name.startswith("mgp::fused_"), name
    if cid == 5:  # the prediction variant of the rhs-column kernel (the round-3 parity gap)
        assert name.startswith(("mgp::fused_rhs_mf_kernel<16", "mgp::fused_rhs_kernel<float,16,true")), name


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_every_kernel_the_default_bench_line_times_is_oracle_checked():
    """The driver's command (`python bench.py`, shortened loops): every `kernel` in the line -- headline and every
    secondary -- must be an instantiation the oracle comparison above has run, and every in-bench check must hold."""
    out = _run([sys.executable, BENCH, "--steps", "6", "--warmup", "3", "--cpu-sample", "0"])
    checked = {_oracle_checked_kernel(c) for c in (2, 3, 4, 5)}
    checked |= {_oracle_checked_kernel(2, "dropin"), _oracle_checked_kernel(2, "dropin_plain")}
    assert out["check"]["ok"], out["check"]
    assert out["roofline"]["kernel"] in checked, (out["roofline"]["kernel"], checked)
    assert set(out["secondary"]) >= {"dropin", "dropin_plain", "c3", "c4", "c5", "knn", "points8M", "c3_shard8", "c4_shard8", "c4_grad", "c2_bwd"}
    for name, sec in out["secondary"].items():
        assert "error" not in sec, (name, sec)
        if name not in ("c4_grad", "c2_bwd"):  # (backward instantiations: checked in the line -- finite differences / fp64)
            assert sec["kernel"] in checked, (name, sec["kernel"], checked)
        assert sec["check"]["ok"], (name, sec["check"])
        assert np.isfinite(sec["value"]) and sec["value"] > 0
    assert out["secondary"]["c3_shard8"]["batch"] == 125_000
    # config 4 at the largest single-GPU size (b = N = 10 M) and its eighth with one acquisition step of the Bayes loop
    assert out["secondary"]["c4"]["batch"] == 10_000_000 and out["secondary"]["c4_shard8"]["batch"] == 1_250_000
    # ... and one evaluation WITH its analytic gradient (round 6: the backward on the dealt-triangle forward kernel)
    assert "backward" in out["secondary"]["c4_grad"]["kernel"], out["secondary"]["c4_grad"]["kernel"]
    # ... and the full backward at the headline shape (feature cotangents included) on the forward kernel
    assert "backward" in out["secondary"]["c2_bwd"]["kernel"], out["secondary"]["c2_bwd"]["kernel"]
    assert 0.0 < out["secondary"]["c4_shard8"]["acquisition_ms"] < 60.0
    # the record the driver keeps (`roofline` survives its parser, unknown top-level keys do not) holds all of it
    compact = out["roofline"]["secondaries"]
    assert set(compact) == set(out["secondary"])
    for name, c in compact.items():
        assert c["value"] == out["secondary"][name]["value"] and c["check_ok"] is True and c["kernel"], (name, c)
    assert "rate_vs_full_batch" in compact["c3_shard8"] and "rate_vs_full_batch" in compact["c4_shard8"]
    assert out["roofline"]["ramp_value"] == out["ramp"]["value"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SMALL = ["--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-secondary", "--points", "60000", "--batch", "20001",
         "--one-device", "--backend", "gloo"]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
@pytest.mark.parametrize("config", ["2", "3"])
def test_bench_two_ranks_self_launched(scaling, config):
    """`python bench.py --gpus 2`: the parent starts the two ranks (fresh children, no re-exec); both take part in
    the collectives; weak scaling doubles the work, strong scaling shards one batch with the reference's chunk rule."""
    out = _run([sys.executable, BENCH, "--gpus", "2", "--config", config, "--scaling", scaling, *SMALL])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["collective_backend"] == "gloo"
    assert out["scaling"] == scaling
    assert np.isfinite(out["value"]) and out["value"] > 0 and out["check"]["ok"]
    if scaling == "weak":
        assert out["config"]["batch_per_gpu"] == 20001 and out["config"]["batch_all_gpus"] == 40002
    else:  # 20001 rows over 2 ranks: floor to rank 0, the remainder to the last rank (_src/mpi_utils.py:36-41)
        assert out["config"]["batch_per_gpu"] == 10000 and out["config"]["batch_all_gpus"] == 20001
    assert abs(out["value"] - out["config"]["batch_all_gpus"] / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]


def test_bench_two_ranks_under_torch_distributed_run():
    """The driver's multi-GPU command line (ranks from the environment)."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2", *SMALL])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0 and out["check"]["ok"]


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", *SMALL], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
