mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in cnt0 cnt1 cur; do
  if [ $v = cur ]; then unset IPP_HIP_LIB; else export IPP_HIP_LIB=$PWD/tools/probes/libipp_$v.so; fi
  python bench.py --no-extra --no-cpu-baseline --steps 80 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'])"
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab/counters.txt
