mkdir -p gpurun_out/r03
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -6
python3 bench.py --cpu-sample 0 --steps 10 > gpurun_out/r03/bench_sec.json 2> gpurun_out/r03/bench_sec.err; tail -3 gpurun_out/r03/bench_sec.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_sec.json'))
print('headline', round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_with_pack_per_step'), d['roofline']['kernel'])
for k,v in d.get('secondary',{}).items():
    print(k, v.get('error') or (round(v['value']/1e6,1), round(v['ms_per_step'],3), round(v['roofline']['frac'],4), round(v['valu']['frac'],4), v['kernel']))
PY
