#!/bin/bash
# A/B of engine switches on a bench configuration, one box: usage: CFG="--grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4" bash tools/ab_cfg.sh "ENV=VAL ENV2=VAL" "..." ...
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-extra ${CFG} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('[$v]', 'value %.2f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms_avg'], 'frac %.3f' % r['frac'], 'step_frac %.3f' % r['step_frac'], 'single %.4f' % (r['single_launch_ms_avg'] or 0), 'P %.4f' % r['other_kernels_ms_avg']['k_prepare'], 'sync', '%.4f' % (c['sync_schedule'] or {}).get('ms_per_step', 0), 'bad', c['items_with_nonzero_status'], c['non_finite_rewards'])"
  done
done 2>&1 | tee -a $O/ab_cfg.txt
