import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from bench import synth, random_neighbors
from muygpys_amd.fused import KernelSpec, posterior_mean_var, loocv_partials
n = b = 1_000_000; k = 30; d = 40
X, y = synth(n, d, 1, 1)
Xd, yd = torch.from_numpy(X).cuda().float(), torch.from_numpy(y).cuda().float()
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).cuda(), torch.from_numpy(ni).cuda()
spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
def t(fn, reps=12):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[3:]))
print("mean+var          ", t(lambda: posterior_mean_var(spec, Xd, Xd, bi, ni, yd)))
print("mean+var+yk       ", t(lambda: posterior_mean_var(spec, Xd, Xd, bi, ni, yd, want_ykinvy=True)))
print("loocv_partials    ", t(lambda: loocv_partials(spec, Xd, yd, bi, ni)))
print("loocv + tolist    ", t(lambda: loocv_partials(spec, Xd, yd, bi, ni)[0].tolist()))
bi2 = torch.arange(b, device="cuda")
print("mean+var bi=arange", t(lambda: posterior_mean_var(spec, Xd, Xd, bi2, ni, yd)))
