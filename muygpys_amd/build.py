"""Build the HIP shared library in-tree (muygpys_amd/lib/libmuygpys_hip.so) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting .so is git-ignored but travels to the GPU box with the snapshot.
"""

from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmuygpys_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         # explicit vector types carry the packed-f32 math; the SLP vectoriser only shuffles registers
         "-fno-slp-vectorize",
         # every `#pragma unroll` of the kernels is meant: loops over a lane's register groups index registers at
         # compile time or not at all (for every file since round 5 -- tools/mkvariant.sh and the run-time compiler
         # always had it, and the regular build of the fp64 128-slot kernel ran 12-14 % behind its own A/B copy)
         "-mllvm", "-pragma-unroll-threshold=1000000", "-Wno-pass-failed"]
# the k-NN scans test the MFMA results right away: keep them in VGPRs (no v_accvgpr_read per value)
PER_FILE_FLAGS = {
    "mgp_knn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # the 64-step elimination must unroll completely (the fp64 / 16-response body exceeds the default
    # pragma-unroll budget, and a rolled loop indexes the 128-register row at run time = in scratch)
    "mgp_fused_rhs.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
    "mgp_fused_rhs_mf.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                             "-Rpass-analysis=kernel-resource-usage"],
    # (+ the register report of every instantiation -> lib/kernel_resources.json: the headline kernels sit at
    # the 168-register cap of three waves per SIMD, and a spill there costs 30 %)
    "mgp_fused_wave.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
    **{f"mgp_fused_wave_inst_{n}.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-Rpass-analysis=kernel-resource-usage"]
       for n in ("f32", "f64")},
    "mgp_solve_wave.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
    "mgp_fused_wide.hip": ["-Rpass-analysis=kernel-resource-usage"],
    "mgp_fused_wide64.hip": ["-Rpass-analysis=kernel-resource-usage"],
    "mgp_backward.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
    "mgp_backward_wave.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-Rpass-analysis=kernel-resource-usage"],
    "mgp_backward_dlt.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-Rpass-analysis=kernel-resource-usage"],
}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # (this file too: its per-file compiler flags are part of what the library is)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")) + [__file__]
    return any(os.path.getmtime(p) > t for p in deps)


def _headers_of(path: str, seen=None) -> set:
    """The project headers a source file includes, transitively (`#include "..."` under csrc/ and include/)."""
    import re

    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for base in (CSRC, os.path.join(HERE, "..", "include")):
            h = os.path.normpath(os.path.join(base, name))
            if os.path.exists(h) and h not in seen:
                seen.add(h)
                _headers_of(h, seen)
    return seen


def _object_stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src, __file__, *_headers_of(src)])


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the .hip files under csrc/ whose object is older than the source or a header it includes (one object
    per file, in parallel; `force`: all of them) and link the .so."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    resources = {}
    res_path = os.path.join(LIBDIR, "kernel_resources.json")
    if not force and os.path.exists(res_path):  # (the records of the objects that are kept)
        import json

        with open(res_path) as f:
            resources = json.load(f)
    extra = os.environ.get("MGP_EXTRA_HIPCC_FLAGS", "").split()
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and not extra and not _object_stale(src, obj):
            continue
        cmd = [hipcc, *FLAGS, *PER_FILE_FLAGS.get(os.path.basename(src), []), *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out)
        if "-Rpass-analysis=kernel-resource-usage" in cmd:
            resources.update(_parse_resource_remarks(out))
            out = "\n".join(ln for ln in out.splitlines() if "kernel-resource-usage" not in ln and "remark:" not in ln)
        if verbose and out.strip():
            print(out, file=sys.stderr)
    if resources:
        import json

        with open(os.path.join(LIBDIR, "kernel_resources.json"), "w") as f:
            json.dump(resources, f, indent=1, sort_keys=True)
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB + ".tmp", *objs, "-ldl"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + " ".join(link) + "\n" + r.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def _parse_resource_remarks(text: str) -> dict:
    """{mangled kernel name: {"VGPRs": n, "VGPRs Spill": n, ...}} from -Rpass-analysis=kernel-resource-usage."""
    import re

    out, cur = {}, None
    for ln in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\S+) \[-Rpass-analysis", ln)
        if m and cur is not None:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.isdigit() else v
    return out


# Static shapes compiled into the run-time-specialisation cache at build time (csrc/mgp_jit.hip; hiprtc
# needs no GPU): the nn_count x feature_count grid of profiles/r03_shape_sweep.md, prepared and plain
# tables in fp32, prepared tables in fp64.  Any other shape is compiled on first use (~1 s).
PREWARM_K = (10, 20, 25, 30, 40, 50)
PREWARM_D = (8, 16, 32, 40, 64)
# ... and the fp64 hyper-parameter backward of a few shapes next to BASELINE config 4's built-in (50, 8) (round 6)
PREWARM_BWD = ((8, 40, 8), (8, 40, 16), (8, 50, 16), (8, 32, 8), (4, 30, 16), (4, 20, 40), (4, 25, 8), (8, 30, 40), (4, 50, 8), (4, 40, 16), (4, 10, 8), (4, 30, 100))


def _prewarm_one(job):
    import ctypes

    es, k, d, packed = job
    lib = ctypes.CDLL(LIB)
    if packed == "bwd":
        lib.mgp_jit_prepare_backward.argtypes = [ctypes.c_int] * 4
        lib.mgp_jit_prepare_backward.restype = ctypes.c_int
        return job, lib.mgp_jit_prepare_backward(es, k, d, 2)
    lib.mgp_jit_prepare.argtypes = [ctypes.c_int] * 6
    lib.mgp_jit_prepare.restype = ctypes.c_int
    return job, lib.mgp_jit_prepare(es, k, 1, d, packed, 2)  # kernel id 2 = Matern-3/2 (any Gram-form kernel)


def prewarm(verbose: bool = False) -> int:
    """Fill muygpys_amd/lib/jit/ for the common shapes (objects already there are kept: the file name
    carries a hash of the kernel sources).  Returns the number of shapes available."""
    import multiprocessing as mp

    jobs = [(4, k, d, p) for k in PREWARM_K for d in PREWARM_D for p in (1, 0)]
    jobs += [(8, k, d, 1) for k in PREWARM_K for d in PREWARM_D if d <= 32]
    jobs += [(es, k, d, "bwd") for es, k, d in PREWARM_BWD]
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        done = pool.map(_prewarm_one, jobs)
    ok = sum(1 for _, rc in done if rc == 0)
    if verbose:
        print(f"run-time specialisation cache: {ok} of {len(jobs)} shapes ready", file=sys.stderr)
    # drop objects of other builds: a file name ends in the hash of kernel sources + options + compiler, and the
    # library says which one is its own (the newest file need not be: an A -> B -> A source sequence keeps A's
    # older objects, and a first-use compile of another source tree may be newer still)
    jit = os.path.join(LIBDIR, "jit")
    if ok and os.path.isdir(jit):
        import ctypes

        lib = ctypes.CDLL(LIB)
        buf = ctypes.create_string_buffer(32)
        if lib.mgp_jit_source_hash(buf, 32) == 0:
            current = buf.value.decode() + ".hsaco"
            for f in os.listdir(jit):
                if f.endswith(".hsaco") and f.rsplit("_", 1)[-1] != current:
                    os.remove(os.path.join(jit, f))
    return ok


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    prewarm(verbose=True)
