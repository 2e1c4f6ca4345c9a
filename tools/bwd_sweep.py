#!/usr/bin/env python3
"""fp64 hyper-parameter backward, shape by shape: the dealt-triangle forward kernel's BWD instantiation (round 6, built-in
or compiled at run time) against round 5's row-per-lane kernel (MGP_BACKWARD_DLT=0), 1 M neighbourhoods each.

    python tools/bwd_sweep.py [--md profiles/r06_backward_sweep_f64.md]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(32, 8), (40, 8), (50, 8), (62, 8), (40, 16), (50, 16), (62, 16), (40, 32), (50, 32)]


def one(k, d, dlt):
    env = dict(os.environ, MUYGPYS_HIP_JIT="force")
    if not dlt:
        env["MGP_BACKWARD_DLT"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gradbench.py"), "--k", str(k), "--d", str(d), "--b", "1000000",
                        "--n", "4000000"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
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
    a = ap.parse_args()
    lines = ["# fp64 hyper-parameter backward by shape (1 M neighbourhoods, anisotropic Matern-3/2, 4 M-row table)", "",
             "| nn_count | features | forward ms | backward ms (round 6: forward kernel, BWD) | backward ms (round 5: row per lane) | ratio |",
             "|---|---|---|---|---|---|"]
    for k, d in SHAPES:
        new, old = one(k, d, True), one(k, d, False)
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
