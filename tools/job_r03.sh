python3 -m pytest tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -12
