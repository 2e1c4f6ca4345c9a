mkdir -p gpurun_out/r03
python3 tools/abtime.py --variants vC,default --rounds 3 > gpurun_out/r03/ab5.log 2>&1
cat gpurun_out/r03/ab5.log
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -8
python3 bench.py --cpu-sample 0 --steps 10 2>&1 | tail -1 > gpurun_out/r03/bench_c2.json; cat gpurun_out/r03/bench_c2.json | cut -c1-600
