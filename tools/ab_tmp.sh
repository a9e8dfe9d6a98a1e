timeout 900 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "device_search" -s 2>&1 | tail -12
