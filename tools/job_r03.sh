python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3
python3 bench.py --cpu-sample 0 --no-secondary --config 5 --steps 5 | python3 -c "
import json,sys; d=json.load(sys.stdin); print('c5', round(d['value']/1e6,1),'M/s', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
python3 tools/abtime.py --variants default --rounds 2 --aniso 1
python3 tools/abtime.py --variants default --rounds 2 --dtype f64
