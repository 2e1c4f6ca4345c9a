mkdir -p gpurun_out/r02
for v in 0 1; do
  if [ $v = 1 ]; then export MGP_GRAM=1; fi
  python3 bench.py --cpu-sample 0 > gpurun_out/r02/bench_gram$v.json 2> gpurun_out/r02/bench_gram$v.err
  python3 - <<PY
import json
d=json.load(open('gpurun_out/r02/bench_gram$v.json'))
print('gram=$v', round(d['value']/1e6,1),'M/s', d['roofline']['kernel_ms'], d['roofline']['frac'])
PY
done
python3 - <<'PY'
import os, subprocess, sys
code = '''
import torch, numpy as np, sys
sys.path.insert(0, ".")
from bench import synth, random_neighbors
from muygpys_amd.fused import KernelSpec, posterior_mean_var
n,b,k,d=200000,200000,30,40
X,y=synth(n,d,1)
Xd=torch.from_numpy(X).cuda(); yd=torch.from_numpy(y).cuda()
bi,ni=random_neighbors(n,b,k,2); bi=torch.from_numpy(bi).cuda(); ni=torch.from_numpy(ni).cuda()
m,v=posterior_mean_var(KernelSpec("matern15","l2",5.0,1e-3),Xd,Xd,bi,ni,yd,packed=True)
m64,v64=posterior_mean_var(KernelSpec("matern15","l2",5.0,1e-3),Xd.double(),Xd.double(),bi,ni,yd.double(),packed=False)
torch.cuda.synchronize()
print("mean err", float((m.double()-m64).abs().max()), "rms", float(m64.pow(2).mean().sqrt()), "var relerr", float(((v.double()-v64)/v64).abs().max()))
'''
for g in ("", "1"):
    env = dict(os.environ)
    env.pop("MGP_GRAM", None)
    if g: env["MGP_GRAM"] = "1"
    print("MGP_GRAM=", g, subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout)
PY
