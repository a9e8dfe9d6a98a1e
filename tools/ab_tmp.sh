timeout 600 python -m pytest tests/test_hip_mcts.py -x -q -m gpu -k "device_search" 2>&1 | tail -30
PYTHONPATH=$PWD python tools/mcts_bench.py --reps 3
