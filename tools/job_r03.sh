python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_backend.py -m gpu -q 2>&1 | tail -2
python3 bench.py --cpu-sample 0 --no-secondary --config 5 --steps 30 --warmup 10 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c5', round(d['value']/1e6,1),'M/s', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
