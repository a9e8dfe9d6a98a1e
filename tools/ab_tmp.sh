python -m pytest tests/test_hip_tree.py tests/test_hip_mcts.py tests/test_hip_parity.py tests/test_hip_window.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err; python - <<'PY'
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['roofline']['frac'], d['ms_per_step'])
for e in d['extra']: print(e['name'][:60], '|', round(e['value']), e.get('kernel_ms_avg'), e.get('frac'))
PY
