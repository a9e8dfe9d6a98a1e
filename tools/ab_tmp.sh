python -m pytest tests/test_hip_mcts.py -x -q 2>&1 | tail -4
python - <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
import bench
r = bench.run_mcts_driver(torch, "cuda:0"); print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "name"})
PY
