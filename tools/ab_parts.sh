#!/bin/bash
# Schedules of a batched step on one box: one launch per step (parts 1) against 2 / 4 groups on their own streams.
# usage (on the GPU box): [BENCH_ARGS="--envs 32768"] bash tools/ab_parts.sh [parts ...]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O
PARTS=("$@"); [ ${#PARTS[@]} -eq 0 ] && PARTS=(1 2 4)
for rep in 1 2; do
  for p in "${PARTS[@]}"; do
    python bench.py --no-cpu-baseline --no-extra --parts $p ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
s=c.get('sync_schedule') or {}
print('parts $p', 'value %.2f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], 'regions', ['%.4f' % x for x in c['region_ms_per_step']],
      'kernel_ms/step %.4f' % r['kernel_ms_avg'], 'single %.4f' % (r['single_launch_ms_avg'] or 0), 'part %.4f' % (r.get('part_launch_ms_avg') or 0),
      'frac %.3f' % r['frac'], 'sync %.2f M' % (s.get('value', 0)/1e6), 'bad', c['items_with_nonzero_status'], c['non_finite_rewards'])
"
  done
done 2>&1 | tee -a $O/ab_parts.txt
