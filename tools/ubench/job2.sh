set -x
mkdir -p gpurun_out/r02
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1
tail -5 gpurun_out/r02/pytest_gpu.log
python3 bench.py --cpu-sample 0 > gpurun_out/r02/bench_packed.json 2> gpurun_out/r02/bench_packed.err
python3 bench.py --cpu-sample 0 --no-prepared-tables > gpurun_out/r02/bench_plain.json 2> gpurun_out/r02/bench_plain.err
python3 bench.py --cpu-sample 0 --config 3 > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err
python3 bench.py --cpu-sample 0 --config 4 --steps 5 > gpurun_out/r02/bench_c4.json 2> gpurun_out/r02/bench_c4.err
python3 bench.py --cpu-sample 0 --config 5 --steps 5 > gpurun_out/r02/bench_c5.json 2> gpurun_out/r02/bench_c5.err
python3 bench.py --cpu-sample 0 --gpus 2 --one-device --backend gloo --steps 5 --batch 200000 > gpurun_out/r02/bench_2rank.json 2> gpurun_out/r02/bench_2rank.err
python3 bench.py --gpus 2 --steps 2 > gpurun_out/r02/bench_2gpu_fail.json 2> gpurun_out/r02/bench_2gpu_fail.err; echo "rc=$?" >> gpurun_out/r02/bench_2gpu_fail.err
cat gpurun_out/r02/bench_*.json
