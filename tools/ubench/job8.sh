python3 -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "wide" 2>&1 | tail -5
for kk in 70 80 100 126; do python3 tools/kbench.py --k $kk --d 40 --b 200000 --paths auto,generic --rounds 3 2>&1 | tail -2; done
python3 tools/kbench.py --k 100 --d 8 --b 200000 --paths auto --rounds 3 2>&1 | tail -1
