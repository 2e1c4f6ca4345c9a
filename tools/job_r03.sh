python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_backend.py tests/test_gpu_functor_layer.py -m gpu -q -x 2>&1 | tail -3
for v in variants/lib_rhsfwd.so muygpys_amd/lib/libmuygpys_hip.so; do
MUYGPYS_HIP_LIB=$PWD/$v python3 bench.py --cpu-sample 0 --no-secondary --config 5 --steps 5 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c5', round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu']['frac'])"
done
python3 tools/kbench.py --k 64 --d 40 --R 16 --b 300000 --kernel rbf --metric F2 --paths auto,generic --packed 0 --rounds 3 2>&1 | tail -2
python3 tools/kbench.py --k 40 --d 16 --R 8 --b 300000 --dtype f64 --paths auto,generic --packed 0 --rounds 3 2>&1 | tail -2
