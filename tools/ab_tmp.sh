mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2b/pytest.log 2>&1; tail -14 gpurun_out/r2b/pytest.log
python tools/compat_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2b/compat.txt
python bench.py > gpurun_out/r2b/bench.log 2>&1; tail -c 5000 gpurun_out/r2b/bench.log
