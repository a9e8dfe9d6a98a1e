mkdir -p gpurun_out/ab
C2="--grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.4f ms/step' % d['ms_per_step'], r['kernel'], 'kernel %.4f ms' % r['kernel_ms_avg'], '%.0f GB/s' % r['achieved'], 'prep %.3f' % r['other_kernels_ms_avg']['k_prepare'], 'bad', d['config']['items_with_nonzero_status'], d['config']['non_finite_rewards'])"; }
{
timeout 900 python -m pytest tests -m gpu -x -q -rs 2>&1 | tail -4
python bench.py --no-extra --no-cpu-baseline 2>/dev/null | show "cfg1"
python bench.py --no-extra --no-cpu-baseline $C2 2>/dev/null| show "cfg2 auto"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab/exp8.txt
