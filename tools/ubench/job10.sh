python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_properties.py tests/test_gpu_backend.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2 3; do python3 bench.py --cpu-sample 0 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c2', round(d['value']/1e6,1),'M/s', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
