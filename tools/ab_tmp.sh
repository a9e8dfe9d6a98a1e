python -m pytest tests/test_hip_classes.py tests/test_hip_features.py tests/test_hip_sharding.py -x -q 2>&1 | tail -15
python tools/compat_bench.py 2>&1 | grep -v amdgpu
