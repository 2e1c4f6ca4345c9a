python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_jit.py tests/test_gpu_backend.py -m gpu -q 2>&1 | tail -2
MUYGPYS_HIP_JIT=0 python3 tools/kbench.py --k 50 --d 16 --R 1 --b 200000 --dtype f64 --paths auto --packed 1 --rounds 3 2>&1 | tail -1
MUYGPYS_HIP_JIT=0 python3 tools/kbench.py --k 25 --d 16 --R 1 --b 500000 --paths auto --packed 1 --rounds 3 2>&1 | tail -1
python3 tools/abtime.py --variants default --rounds 2 --iters 60
