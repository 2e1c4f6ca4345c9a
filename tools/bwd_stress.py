#!/usr/bin/env python3
"""Backward instantiations of the forward kernel against round 5's kernels (MGP_BACKWARD_DLT=0, a child process) on random
shapes, noise models and batch sizes -- every cotangent (GPU box): python tools/bwd_stress.py [--cases 24]"""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run_case(case, out):
    import torch

    from muygpys_amd import _lib

    k, d, R, kid, aniso, nmode, b, es = case
    td = torch.float32 if es == 4 else torch.float64
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    n = 4000
    dev = "cuda"
    X = torch.from_numpy(rng.normal(size=(n, d))).to(dev, td)
    Y = torch.from_numpy(np.sin(rng.normal(size=(n, R)))).to(dev, td)
    bi = torch.from_numpy(rng.choice(n, size=b, replace=False)).to(dev)
    ni = torch.from_numpy(np.stack([rng.choice(n - 1, size=k, replace=False) for _ in range(b)])).to(dev)
    ni = ni + (ni >= bi[:, None])
    ls = torch.from_numpy(np.sqrt(d) * rng.uniform(0.7, 1.5, size=d if aniso else 1)).to(dev, td)
    gm = torch.from_numpy(rng.normal(size=(b, R))).to(dev, td)
    gv = torch.from_numpy(rng.normal(size=b)).to(dev, td)
    nd = None
    if nmode == 1:
        nd = torch.from_numpy(rng.uniform(1e-2, 5e-2, size=n)).to(dev, td)
    elif nmode == 2:
        nd = torch.from_numpy(rng.uniform(1e-2, 5e-2, size=(b, k))).to(dev, td)
    gx, gy = torch.zeros_like(X), torch.zeros_like(Y)
    gl = torch.zeros((b, ls.numel()), device=dev, dtype=td)
    gn = torch.zeros((b, k), device=dev, dtype=td)
    info = torch.zeros(1, device=dev, dtype=torch.int32)
    P = _lib.ptr
    _lib.load().mgp_jit_prepare_backward(es, k, d, kid)
    rc = _lib.fn("posterior_backward", td)(P(X), P(X), d, P(bi), P(ni), b, k, P(Y), R, nmode, 2e-2, P(nd) if nd is not None else None,
                                           kid, 0, P(ls), ls.numel(), P(gm), P(gv), P(gx), P(gx), P(gy), P(gl), P(gn), P(info),
                                           _lib.stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    np.savez(out, gx=gx.double().cpu().numpy(), gy=gy.double().cpu().numpy(), gl=gl.double().sum(0).cpu().numpy(),
             gn=gn.double().cpu().numpy(), kernel=np.array(_lib.last_kernel()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    cases = []
    for _ in range(args.cases):
        es = 4 if rng.random() < 0.75 else 8
        k = int(rng.integers(3, 31))
        d = int(rng.choice([36, 40, 40, 40, 32, 16, 8, 64] if es == 4 else [8, 16, 40]))
        cases.append((k, d, int(rng.choice([1, 1, 1, 3])), int(rng.choice([0, 2, 3, 4])), bool(rng.integers(0, 2)), int(rng.integers(0, 3)),
                      int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 4001 % 3999])), es))
    if args.child:
        idx, out = args.child.split(":")
        run_case(cases[int(idx)], out)
        return
    bad = 0
    for idx, case in enumerate(cases):
        outs = []
        for env in ({}, {"MGP_BACKWARD_DLT": "0"}):
            out = f"/tmp/bwd_stress_{idx}_{len(outs)}.npz"
            r = subprocess.run([sys.executable, __file__, "--cases", str(args.cases), "--child", f"{idx}:{out}"], env=dict(os.environ, **env),
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(case, "FAILED", r.stderr[-400:])
                bad += 1
                break
            outs.append(np.load(out))
        if len(outs) < 2:
            continue
        a, c = outs
        tol = 3e-3 if case[-1] == 4 else 1e-8
        errs = {key: float(np.abs(a[key] - c[key]).max() / (np.abs(c[key]).max() + 1e-300)) for key in ("gx", "gy", "gl", "gn")}
        ok = all(v <= tol for v in errs.values())
        bad += 0 if ok else 1
        print(case, "ok" if ok else "MISMATCH", {k_: f"{v:.1e}" for k_, v in errs.items()}, str(a["kernel"])[-60:], "|", str(c["kernel"])[-40:], flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
