python3 -m pytest tests/test_gpu_backward.py tests/test_oracle_grad_golden.py -m gpu -q 2>&1 | tail -2
python3 tools/bwdbench.py --outs "x,ls,noise,y;ls,noise;x" 2>&1 | tail -4
