python3 tools/abtime.py --variants default --rounds 3
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3
python3 bench.py > gpurun_out/bench_fold2.json 2> gpurun_out/bench_fold2.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_fold2.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(v.get('value'),v.get('roofline',{}).get('frac')) for k,v in d['secondary'].items()})"
