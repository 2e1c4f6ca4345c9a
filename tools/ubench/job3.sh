set -x
mkdir -p gpurun_out/r02
python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_fused.py -m gpu -x -q > gpurun_out/r02/pytest_gpu2.log 2>&1
tail -5 gpurun_out/r02/pytest_gpu2.log
python3 bench.py --cpu-sample 0 > gpurun_out/r02/bench_packed.json 2> gpurun_out/r02/bench_packed.err
python3 bench.py --cpu-sample 0 --no-prepared-tables > gpurun_out/r02/bench_plain.json 2> gpurun_out/r02/bench_plain.err
python3 bench.py --cpu-sample 0 --config 3 > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err
python3 bench.py --cpu-sample 0 --config 4 --steps 5 > gpurun_out/r02/bench_c4.json 2> gpurun_out/r02/bench_c4.err
python3 bench.py --cpu-sample 0 --config 5 --steps 5 > gpurun_out/r02/bench_c5.json 2> gpurun_out/r02/bench_c5.err
for f in packed plain c3 c4 c5; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/r02/bench_$f.json'))
print('$f', round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['valu']['achieved'])"; done
