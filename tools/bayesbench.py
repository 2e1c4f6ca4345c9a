#!/usr/bin/env python3
"""Host / device time of one acquisition step of the Bayes driver, by part (round 6):
    python tools/bayesbench.py [dims] [recorded trials]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.optimize import minimize

from muygpys_amd._src.optimize.chassis.hip import _UCBBayesOpt

p = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.RandomState(5)
bounds = np.array([[0.1, 10.0]] * p)
opt = _UCBBayesOpt(lambda **kw: 0.0, [f"l{i}" for i in range(p)], bounds, random_state=7)
for x in opt._sample(n):
    opt.X.append(x)
    opt.y.append(-float(((np.log(x) - 0.3) ** 2).sum()) + 0.01 * rng.randn())


def clock(fn, reps=30):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, np.mean(ts) * 1e3


gp = opt.gp
print("fit                    median %.3f ms  mean %.3f" % clock(lambda: gp.fit(np.array(opt.X), np.array(opt.y))))
gen = opt._generator()
print("top_candidates         median %.3f ms  mean %.3f" % clock(lambda: gp.top_candidates(10000, 10, bounds, 2.576, gen)))
seeds, _ = gp.top_candidates(10000, 10, bounds, 2.576, gen)
m = seeds.shape[0]


def polish():
    def neg(z):
        v, g = gp.ucb(z.reshape(m, p), 2.576, want_grad=True)
        return -float(v.sum()), -g.reshape(-1)

    return minimize(neg, seeds.reshape(-1), jac=True, bounds=np.tile(bounds, (m, 1)), method="L-BFGS-B", options={"maxfun": 30})


print("polish (stacked)       median %.3f ms  mean %.3f   nfev %d" % (*clock(polish), polish().nfev))
print("ucb + gradient, once   median %.3f ms  mean %.3f" % clock(lambda: gp.ucb(seeds, 2.576, want_grad=True)))
print("_suggest               median %.3f ms  mean %.3f" % clock(lambda: opt._suggest(2.576)))
