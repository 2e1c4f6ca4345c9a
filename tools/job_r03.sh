python3 -m pytest tests/test_gpu_functor_layer.py -m gpu -q -x -k "difference_isotropy" 2>&1 | tail -8
