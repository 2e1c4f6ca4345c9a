mkdir -p gpurun_out/r02
python3 -m pytest tests/test_gpu_neighbors.py tests/test_gpu_fast_mean.py tests/test_gpu_backward.py tests/test_gpu_fused.py tests/test_gpu_distributed.py -m gpu -q > gpurun_out/r02/pytest_gpu3.log 2>&1
tail -40 gpurun_out/r02/pytest_gpu3.log
python3 bench.py --cpu-sample 8192 > gpurun_out/r02/bench_cpu.json 2> gpurun_out/r02/bench_cpu.err; tail -c 1500 gpurun_out/r02/bench_cpu.json; tail -3 gpurun_out/r02/bench_cpu.err
