#!/usr/bin/env python3
"""Build-parameter sweeps without rebuilding the library (GPU box): every variant is the kernel header set copied to a
scratch directory with some `#define MGP_X default` lines rewritten, compiled by the run-time compiler
(MUYGPYS_HIP_SRC / MUYGPYS_HIP_JIT_CACHE / MUYGPYS_HIP_JIT=force) for a shape that is NOT built into the library, and
timed by tools/kbench.py in a child process.

    python tools/jit_sweep.py --k 29 --d 40 --dtype f32 --variants "base;MGP_CHOL_PRIO=3;MGP_DIST_PRIO=0,MGP_XCHG_PRIO=1"
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muygpys_amd", "csrc")


def make_variant(spec: str, top: str) -> str:
    d = os.path.join(top, re.sub(r"[^A-Za-z0-9_=]+", "_", spec))
    os.makedirs(os.path.join(d, "jit"), exist_ok=True)
    src = os.path.join(d, "csrc")
    os.makedirs(src, exist_ok=True)
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            shutil.copy(os.path.join(CSRC, f), src)
    os.makedirs(os.path.join(d, "include"), exist_ok=True)  # (mgp_device.h includes ../../include/muygpys_hip.h)
    shutil.copy(os.path.join(ROOT, "include", "muygpys_hip.h"), os.path.join(d, "include"))
    if spec != "base":
        for item in spec.split(","):
            name, val = item.split("=")
            hit = 0
            for f in os.listdir(src):
                p = os.path.join(src, f)
                text = open(p).read()
                new, n = re.subn(rf"(#define {name} )(-?[0-9A-Za-z_.]+)", rf"\g<1>{val}", text)
                if n:
                    open(p, "w").write(new)
                    hit += n
            if not hit:
                raise SystemExit(f"{name}: no such #define")
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True)
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--tool", default="kbench.py", help="script under tools/ that does the timing (kbench.py, bwdbench.py)")
    ap.add_argument("--grep", default="path=", help="prefix of the tool's result line(s)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="arguments of tools/kbench.py (after --)")
    args = ap.parse_args()
    rest = [r for r in args.rest if r != "--"]
    top = tempfile.mkdtemp(prefix="mgp_sweep_")
    specs = args.variants.split(";")
    dirs = {s: make_variant(s, top) for s in specs}
    for rep in range(args.repeat):
        for s in specs:
            d = dirs[s]
            # the library resolves ../../include relative to the source directory: csrc/../../include -> d/../include
            env = dict(os.environ, MUYGPYS_HIP_SRC=os.path.join(d, "csrc"), MUYGPYS_HIP_JIT_CACHE=os.path.join(d, "jit"),
                       MUYGPYS_HIP_JIT="force")
            inc = os.path.join(os.path.dirname(d), "include")
            if not os.path.exists(inc):
                shutil.copytree(os.path.join(d, "include"), inc)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", args.tool), "--rounds", str(args.rounds)] + rest,
                               env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith(args.grep)]
            print(f"{s:60s} {' | '.join(line) if line else 'FAILED: ' + r.stderr[-300:]}", flush=True)
    shutil.rmtree(top, ignore_errors=True)


if __name__ == "__main__":
    main()
