python3 -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "rhs" 2>&1 | tail -2
for v in variants/lib_exp1.so muygpys_amd/lib/libmuygpys_hip.so; do
MUYGPYS_HIP_LIB=$PWD/$v python3 bench.py --cpu-sample 0 --no-secondary --config 5 --steps 5 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c5', round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
done
