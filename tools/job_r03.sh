python3 -m pytest tests/test_gpu_matern_gen.py -m gpu -q -x 2>&1 | tail -15
python3 -m pytest tests/test_gpu_functor_layer.py tests/test_gpu_backend.py -m gpu -q -x 2>&1 | tail -4
