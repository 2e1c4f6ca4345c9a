#!/usr/bin/env python3
"""Where one analytic-gradient LOOCV evaluation spends its time (GPU box):
python tools/gradbench.py [--b 2000000 --n 10000000 --k 50 --d 8 --dtype f64 --aniso 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, loocv_partials, loocv_value_and_grad, _length_scale_tensor
ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=2_000_000); ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=50); ap.add_argument("--d", type=int, default=8)
ap.add_argument("--dtype", default="f64"); ap.add_argument("--aniso", type=int, default=1)
a = ap.parse_args()
td = torch.float64 if a.dtype == "f64" else torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((a.n, a.d), device="cuda", dtype=td, generator=g)
y = torch.sin(X[:, 0]) + 0.1 * torch.randn((a.n,), device="cuda", dtype=td, generator=g)
bi = torch.randperm(a.n, device="cuda", generator=g)[: a.b].contiguous()
ni = torch.randint(0, a.n - 1, (a.b, a.k), device="cuda", generator=g)
ni = ni + (ni >= bi[:, None])
ls = [1.5] * a.d if a.aniso else 1.5
spec = KernelSpec("matern15", "l2", ls, 1e-3)
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("forward (loocv_partials)       %.2f ms" % wall(lambda: loocv_partials(spec, X, y, bi, ni)))
print("value_and_grad                 %.2f ms" % wall(lambda: loocv_value_and_grad(spec, X, y, bi, ni)))
gm = torch.randn(a.b, device="cuda", dtype=td); gv = torch.randn(a.b, device="cuda", dtype=td); gy = torch.full((a.b,), 0.01, device="cuda", dtype=td)
lst = _length_scale_tensor(spec.length_scale, a.d, X)
g_l = torch.zeros((a.b, lst.numel()), device="cuda", dtype=td); g_n = torch.zeros((a.b, a.k), device="cuda", dtype=td)
info = torch.zeros(1, device="cuda", dtype=torch.int32); tg = y.reshape(-1, 1).contiguous()
def bwd(gn=True, gl=True):
    rc = _lib.fn("loocv_backward", td)(_lib.ptr(X), a.d, _lib.ptr(bi), _lib.ptr(ni), a.b, a.k, _lib.ptr(tg), 0, 1e-3, None, spec.kernel_id(), spec.metric_id(),
        _lib.ptr(lst), lst.numel(), _lib.ptr(gm), _lib.ptr(gv), _lib.ptr(gy), _lib.ptr(g_l) if gl else None, _lib.ptr(g_n) if gn else None, _lib.ptr(info), _lib.stream_ptr())
    assert rc == 0, rc
print("backward kernel (ls + noise)   %.2f ms" % wall(bwd))
print("backward kernel (ls only)      %.2f ms" % wall(lambda: bwd(gn=False)))
print("column sums of grad_noise      %.2f ms" % wall(lambda: _lib.column_sums(g_n.reshape(-1, 1))))
print("column sums of grad_ls         %.2f ms" % wall(lambda: _lib.column_sums(g_l)))
print("zeros (b,k)                    %.2f ms" % wall(lambda: torch.zeros((a.b, a.k), device="cuda", dtype=td)))
