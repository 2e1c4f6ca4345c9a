python3 -m pytest tests/test_gpu_jit.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -3
python3 tools/shape_sweep.py --ks 40,50 --ds 8,16,32,40,64 2>&1 | grep "^{" | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print(d['k'],d['d'],round(d['ms'],3),round(d['mnbhd'],1))"
MUYGPYS_HIP_JIT=0 python3 tools/shape_sweep.py --ks 40 --ds 8,40 2>&1 | grep "^{" | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print('nojit', d['k'],d['d'],round(d['ms'],3),round(d['mnbhd'],1))"
