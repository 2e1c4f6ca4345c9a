import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from bench import synth, random_neighbors
from muygpys_amd import _lib
from muygpys_amd.fused import KernelSpec, posterior_mean_var
lib = _lib.load()
n = b = 1_000_000; k = 30; d = 40
X, y = synth(n, d, 1, 1)
Xd, yd = torch.from_numpy(X).cuda().float(), torch.from_numpy(y).cuda().float()
bi, ni = random_neighbors(n, b, k, 1)
bi, ni = torch.from_numpy(bi).cuda(), torch.from_numpy(ni).cuda()
spec = KernelSpec("matern15", "l2", 5.0, 1e-3)
mean = torch.empty((b, 1), device="cuda"); var = torch.empty((b,), device="cuda")
import os
TD = torch.float64 if os.environ.get("OCC_F64") else torch.float32
Xd, yd = Xd.to(TD), yd.to(TD); mean = mean.to(TD); var = var.to(TD)
for per_cu in [int(v) for v in os.environ.get("OCC_LIST", "12,11,10,9,8,6,12").split(",")]:
    lib.mgp_debug_set_grid_per_cu(per_cu)
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); posterior_mean_var(spec, Xd, Xd, bi, ni, yd, out_mean=mean, out_var=var); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(per_cu, "WG/CU:", round(float(np.median(ts[2:])), 4), "ms")
