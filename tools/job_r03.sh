python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac']); print({k:(round(v.get('value')/1e6,1),round(v.get('roofline',{}).get('frac'),4), v['steps']) for k,v in d['secondary'].items()}); print(d['cpu_baseline'])"
