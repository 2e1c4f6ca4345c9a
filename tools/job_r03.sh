python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_jit.py -m gpu -q 2>&1 | tail -2
python3 bench.py --cpu-sample 0 --no-secondary --config 4 --steps 8 --warmup 3 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c4', round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
MUYGPYS_HIP_JIT=0 python3 tools/abtime.py --variants default --rounds 2 --dtype f64 --iters 30
