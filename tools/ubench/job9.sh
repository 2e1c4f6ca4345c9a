python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_functor_layer.py tests/test_gpu_fast_mean.py tests/test_gpu_backend.py -m gpu -q -x 2>&1 | tail -5
python3 bench.py --cpu-sample 0 --config 4 --steps 5 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c4', round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu']['achieved'])"
python3 tools/kbench.py --k 50 --d 8 --dtype f64 --aniso 1 --b 1000000 --paths auto --packed 0,1 --rounds 3 | tail -2
python3 tools/kbench.py --k 50 --d 8 --dtype f32 --b 1000000 --paths auto --rounds 3 | tail -1
python3 tools/kbench.py --k 60 --d 40 --dtype f32 --b 1000000 --paths auto --rounds 3 | tail -1
python3 tools/kbench.py --k 40 --d 40 --dtype f64 --b 500000 --paths auto --rounds 3 | tail -1
