#!/usr/bin/env python3
"""Hyper-parameter backward, shape by shape: the forward kernel's BWD instantiations (round 6: dealt triangle for fp64
shapes with 33 .. 64 slots, row per lane for 17 .. 32 slots of either type; built-in or compiled at run time) against
round 5's kernels (MGP_BACKWARD_DLT=0), 1 M neighbourhoods each.

    python tools/bwd_sweep.py [--dtype f64|f32] [--md profiles/r06_backward_sweep_f64.md]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"f64": [(32, 8), (40, 8), (50, 8), (62, 8), (40, 16), (50, 16), (62, 16), (40, 32), (50, 32), (20, 8), (30, 40)],
          "f32": [(16, 8), (20, 16), (25, 8), (30, 16), (30, 40), (30, 64), (20, 40)]}


def one(k, d, dlt, dtype="f64"):
    env = dict(os.environ, MUYGPYS_HIP_JIT="force")
    if not dlt:
        env["MGP_BACKWARD_DLT"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gradbench.py"), "--k", str(k), "--d", str(d), "--b", "1000000",
                        "--n", "4000000", "--dtype", dtype], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    vals = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("forward"):
            vals["fwd"] = float(ln.split()[-2])
        if ln.startswith("backward kernel (ls + noise)"):
            vals["bwd"] = float(ln.split()[-2])
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default="")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    a = ap.parse_args()
    lines = [f"# {a.dtype} hyper-parameter backward by shape (1 M neighbourhoods, anisotropic Matern-3/2, 4 M-row table)", "",
             "| nn_count | features | forward ms | backward ms (round 6: forward kernel, BWD) | backward ms (round 5: row per lane) | ratio |",
             "|---|---|---|---|---|---|"]
    for k, d in SHAPES[a.dtype]:
        new, old = one(k, d, True, a.dtype), one(k, d, False, a.dtype)
        if "bwd" in new and "bwd" in old:
            lines.append(f"| {k} | {d} | {new.get('fwd', float('nan')):.2f} | {new['bwd']:.2f} | {old['bwd']:.2f} | {old['bwd'] / new['bwd']:.2f} x |")
        else:
            lines.append(f"| {k} | {d} | - | {new.get('bwd', 'failed')} | {old.get('bwd', 'failed')} | |")
        print(lines[-1], flush=True)
    if a.md:
        with open(a.md, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
