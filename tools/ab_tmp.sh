mkdir -p gpurun_out/ab
C3="--grid 50 --envs 32768 --episode-steps 40 --steps 20 --warmup 4"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.4f ms/step' % d['ms_per_step'], r['kernel'], 'kernel %.4f ms' % r['kernel_ms_avg'], '%.0f GB/s' % r['achieved'])"; }
{
for rep in 1 2 3; do
for v in prev cur; do
  if [ $v = cur ]; then unset IPP_HIP_LIB; else export IPP_HIP_LIB=$PWD/tools/probes/libipp_$v.so; fi
  python bench.py --no-extra --no-cpu-baseline 2>/dev/null | show "cfg1 $v"
  python bench.py --no-extra --no-cpu-baseline $C3 2>/dev/null| show "cfg3 $v"
done; done
export IPP_HIP_LIB=$PWD/tools/probes/libipp_timing.so
IPP_TIMELINE_FILE=/tmp/tl1.bin python bench.py --no-extra --no-cpu-baseline --steps 20 > /dev/null 2>&1
python tools/timeline.py /tmp/tl1.bin 4096 400 | head -12
unset IPP_HIP_LIB
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab/exp14.txt
