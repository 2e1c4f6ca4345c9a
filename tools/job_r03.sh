export MUYGPYS_HIP_JIT=0
python3 tools/abtime.py --variants default,f64fold --rounds 3 --dtype f64
MUYGPYS_HIP_LIB=$PWD/variants/lib_f64fold.so python3 -m pytest tests/test_gpu_fused.py -m gpu -q 2>&1 | tail -2
