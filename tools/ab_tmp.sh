C2="--grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.4f ms/step' % d['ms_per_step'], r['kernel'], 'kernel %.4f ms' % r['kernel_ms_avg'])"; }
for g in 0 1; do
  IPP_GRF_WGLOBAL=$g python tools/grf_bench.py 100:2048 50:102 2>&1 | grep -v amdgpu
  IPP_GRF_WGLOBAL=$g python bench.py --no-extra --no-cpu-baseline $C2 2>/dev/null | show "cfg2 wglobal=$g"
done
IPP_GRF_WGLOBAL=1 python -m pytest tests/test_hip_big_grids.py -q -k grf 2>&1 | tail -2
