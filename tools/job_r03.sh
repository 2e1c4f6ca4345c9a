python3 -m pytest tests/test_gpu_jit.py tests/test_gpu_fused.py tests/test_gpu_backend.py -m gpu -q -x 2>&1 | tail -3
for f in 1 0; do
MGP_TRACE=0 python3 tools/kbench.py --k 28 --d 32 --R 3 --b 500000 --paths auto --packed 1 --rounds 3 2>&1 | tail -1
done
python3 tools/abtime.py --variants default --rounds 2
