"""CPU-side checks of the run-time specialisation (csrc/mgp_jit.hip) and of the pair scheme the static
kernels use (csrc/mgp_fused_wave_kernel.h, wave_dims): no GPU needed -- hiprtc compiles for gfx950
without a device, and the scheme is arithmetic."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PREPARE = r"""
import ctypes, os, sys, time
sys.path.insert(0, %(root)r)
from muygpys_amd import _lib
lib = _lib.load()
assert lib.mgp_jit_mode() == 1
t0 = time.time(); rc = lib.mgp_jit_prepare(4, 17, 2, 24, 1, 2); t1 = time.time()
assert rc == 0, rc
files = sorted(os.listdir(os.environ["MUYGPYS_HIP_JIT_CACHE"]))
assert len(files) == 1 and files[0].startswith("wave_f32_np32_k17_r2_d24_p1_g1_") and files[0].endswith(".hsaco"), files
blob = open(os.path.join(os.environ["MUYGPYS_HIP_JIT_CACHE"], files[0]), "rb").read()
assert blob.startswith(b"MGPJIT1\n_ZN3mgp17fused_wave_kernel") and b"\x7fELF" in blob[:200]
t2 = time.time(); assert lib.mgp_jit_prepare(4, 17, 2, 24, 1, 2) == 0; t3 = time.time()
assert t3 - t2 < 0.5 * (t1 - t0) + 0.05, "the second request must come from the disk cache"
assert lib.mgp_jit_prepare(4, 30, 1, 40, 1, 2) == 0          # built into the library: nothing to compile
assert len(os.listdir(os.environ["MUYGPYS_HIP_JIT_CACHE"])) == 1
assert lib.mgp_jit_prepare(4, 70, 1, 8, 1, 2) == -2           # more than 64 slots
assert lib.mgp_jit_prepare(4, 20, 1, 30, 1, 2) == -2          # rows not 16-byte multiples
assert lib.mgp_jit_prepare(4, 20, 8, 16, 1, 2) == -2          # a prepared table carries at most four fp32 responses
assert lib.mgp_jit_prepare(2, 20, 1, 16, 1, 2) == -1
print("prepared in %%.2f s" %% (t1 - t0))
"""


def test_shape_is_compiled_without_a_gpu_and_cached_on_disk(tmp_path):
    env = dict(os.environ, MUYGPYS_HIP_JIT_CACHE=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    env.pop("MUYGPYS_HIP_JIT", None)
    r = subprocess.run([sys.executable, "-c", PREPARE % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "prepared in" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]


def _wave_dims(np_, k, gram):
    """The pair-scheme part of wave_dims() (csrc/mgp_fused_wave_kernel.h), restated."""
    need = (k + 1) // 2
    ba, bp = 4, (need + 3) // 4
    for c in (3, 5):
        q = (need + c - 1) // c
        if c * q < ba * bp:
            ba, bp = c, q
    if ba * bp + (2 if gram else 1) <= np_ // 2:
        return True, k + 1, ba, bp
    return False, np_, 4, np_ // 8


@pytest.mark.parametrize("gram", [False, True])
def test_every_pair_of_feature_rows_is_computed_by_some_lane(gram):
    """Lane i keeps own rows i + o_j (o_0 = 0, o_j = (j + 1) BP + 1) and reads partner rows i + p, p = 1 .. BP,
    all modulo M: over the lanes 0 .. M-1 the BA x BP pairs per lane must cover every unordered pair of the
    M feature rows, and a pair of a row with itself must never be kept."""
    for k in range(1, 63):
        np_ = 16 if k + 2 <= 16 else (32 if k + 2 <= 32 else 64)
        modm, M, ba, bp = _wave_dims(np_, k, gram)
        own = [0] + [(j + 1) * bp + 1 for j in range(1, ba)]
        seen = set()
        for i in range(M if modm else np_):
            for o in own:
                for p in range(1, bp + 1):
                    r1, c = (i + o) % M, (i + p) % M
                    if r1 != c:
                        seen.add((min(r1, c), max(r1, c)))
        rows = k + 1  # neighbours + query (the generic scheme also visits response / padding slots)
        want = {(a, b) for a in range(rows) for b in range(a + 1, rows)}
        assert want <= seen, (k, np_, modm, ba, bp, sorted(want - seen)[:5])
        if modm:
            assert ba * bp < np_ // 2 and ba * bp >= (k + 1) // 2


def _msgpack_uint_after(blob: bytes, key: str):
    """Value of a small unsigned integer entry of the code object's msgpack metadata (fixstr key)."""
    k = bytes([0xA0 + len(key)]) + key.encode()
    i = blob.find(k)
    if i < 0:
        return None
    j = i + len(k)
    b = blob[j]
    if b < 0x80:
        return b
    width = {0xCC: 1, 0xCD: 2, 0xCE: 4}.get(b)
    return int.from_bytes(blob[j + 1:j + 1 + width], "big") if width else None


def test_prewarmed_instantiations_do_not_spill():
    """The folded elimination parks 48 (fp32, 32 slots) or 96 (64 slots) registers across a task: an
    instantiation that spills because of it runs slower than the unfolded one (measured 2.2 vs 1.64 ms on the
    headline shape).  Every shape compiled into the cache at build time is checked here, without a GPU, from
    the code object's own metadata."""
    from muygpys_amd import build

    build.build()
    assert build.prewarm() > 0
    jit = os.path.join(build.LIBDIR, "jit")
    seen = 0
    for name in sorted(os.listdir(jit)):
        if not name.endswith(".hsaco"):
            continue
        blob = open(os.path.join(jit, name), "rb").read()
        spills = _msgpack_uint_after(blob, ".vgpr_spill_count")
        scratch = _msgpack_uint_after(blob, ".private_segment_fixed_size")
        if "_b1_" in name:
            # backward instantiations (round 6): two waves per SIMD with every squared distance kept through the
            # covariance phase -- a bounded number of spill slots (mostly loop invariants), not none
            assert spills <= 512, (name, spills, scratch)
            continue
        assert spills == 0 and scratch == 0, (name, spills, scratch)
        seen += 1
    assert seen >= 60


def test_jit_sweep_variants_rewrite_the_defaults(tmp_path):
    """tools/jit_sweep.py builds its variants by rewriting `#define MGP_X default` in a copy of the kernel headers: the
    copy must hold every header the run-time compiler hashes, the rewrite must hit exactly the named macro, and an
    unknown name must stop the sweep."""
    import importlib.util
    import re

    import pytest

    from muygpys_amd import build

    spec = importlib.util.spec_from_file_location("jit_sweep", os.path.join(os.path.dirname(build.CSRC), "..", "tools", "jit_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    base = mod.make_variant("base", str(tmp_path))
    var = mod.make_variant("MGP_DLT_CHUNK=6,MGP_CHOL_PRIO=3", str(tmp_path))
    for name in ("mgp_fused_wave_kernel.h", "mgp_wave_common.h", "mgp_args.h", "mgp_device.h", "mgp_loocv_tree.h"):
        assert os.path.exists(os.path.join(base, "csrc", name))
    a = open(os.path.join(base, "csrc", "mgp_fused_wave_kernel.h")).read()
    b = open(os.path.join(var, "csrc", "mgp_fused_wave_kernel.h")).read()
    assert a == open(os.path.join(build.CSRC, "mgp_fused_wave_kernel.h")).read()
    changed = [(x, y) for x, y in zip(a.splitlines(), b.splitlines()) if x != y]
    assert len(changed) == 2 and all(re.match(r"#define MGP_(DLT_CHUNK|CHOL_PRIO) ", y) for _, y in changed), changed
    with pytest.raises(SystemExit):
        mod.make_variant("MGP_NO_SUCH_KNOB=1", str(tmp_path))
