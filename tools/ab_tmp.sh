timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err; python - <<'PY'
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), round(d['roofline']['frac'],4), round(d['ms_per_step'],4))
for e in d['extra']: print(e['name'][:60], '|', round(e['value']), e.get('kernel_ms_avg'), e.get('frac'))
PY
