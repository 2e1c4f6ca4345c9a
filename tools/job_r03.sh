mkdir -p gpurun_out/r03
python3 tools/abtime.py --variants ownreg0,default --rounds 3 > gpurun_out/r03/ab9.log 2>&1
cat gpurun_out/r03/ab9.log
python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_jit.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
