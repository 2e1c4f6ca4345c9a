mkdir -p gpurun_out/r03
python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_jit.py tests/test_gpu_properties.py tests/test_gpu_functor_layer.py -m gpu -q -x 2>&1 | tail -4
python3 tools/abtime.py --variants gram64off,default --k 50 --d 8 --dtype f64 --aniso 1 --rounds 2 --iters 6 > gpurun_out/r03/ab10.log 2>&1
python3 tools/abtime.py --variants gram64off,default --k 30 --d 40 --dtype f64 --rounds 2 --iters 6 >> gpurun_out/r03/ab10.log 2>&1
MUYGPYS_HIP_JIT=0 python3 tools/abtime.py --variants gram64off,default --k 20 --d 16 --dtype f64 --rounds 2 --iters 6 >> gpurun_out/r03/ab10.log 2>&1
cat gpurun_out/r03/ab10.log
