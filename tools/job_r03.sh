python3 -m pytest tests/test_gpu_fused.py -m gpu -q -k "folded" 2>&1 | tail -15
