export MUYGPYS_HIP_JIT=0
python3 tools/abtime.py --variants default,nowait --rounds 3 --iters 60
