mkdir -p gpurun_out/ab
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.4f ms/step' % d['ms_per_step'], r['kernel'], 'kernel %.4f ms' % r['kernel_ms_avg'], '%.0f GB/s' % r['achieved'], 'bad', d['config']['items_with_nonzero_status'], d['config']['non_finite_rewards'])"; }
{
for rep in 1 2; do
for p in 0 1; do
  IPP_PIPE=$p python bench.py --no-extra --no-cpu-baseline 2>/dev/null | show "cfg1 pipe=$p"
done; done
export IPP_HIP_LIB=$PWD/tools/probes/libipp_timing.so
IPP_PIPE=1 IPP_TIMELINE_FILE=/tmp/tl1.bin python bench.py --no-extra --no-cpu-baseline --steps 20 > /dev/null 2>&1
python - <<'PY'
import numpy as np
t=np.fromfile('/tmp/tl1.bin',dtype=np.uint64).reshape(-1,8)[:4096].astype(np.float64)
t0=t[:,0].min()
s,m,e,f=(t[:,0]-t0)/100,(t[:,1]-t0)/100,(t[:,2]-t0)/100,(t[:,6]-t0)/100
sub=[(t[:,3]-t[:,0])/100,(t[:,4]-t[:,3])/100,(t[:,5]-t[:,4])/100,(t[:,1]-t[:,5])/100]
print('producer sub-phases mean us: inputs %.1f obs+mid %.1f gather %.1f solve+Q %.1f'%tuple(x.mean() for x in sub))
print('producer per item: mean %.1f p90 %.1f max %.1f'%((m-s).mean(),np.percentile(m-s,90),(m-s).max()))
print('publish -> first consumer out %.1f, publish -> last consumer out %.1f; span %.1f'%((f-m).mean(),(e-m).mean(),e.max()))
PY
unset IPP_HIP_LIB
python -m pytest tests/test_hip_sharding.py -x -q -k pipelined 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab/exp13.txt
