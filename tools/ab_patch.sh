#!/bin/bash
# A/B of the patch step kernel (waves per item) against the band-tile kernels on the headline workload, one box.
# usage (on the GPU box): [BENCH_ARGS="--envs 32768"] bash tools/ab_patch.sh [name=ENV=VAL,ENV=VAL ...]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$name', 'value %.2f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], r['kernel'], 'kernel_ms %.4f' % r['kernel_ms_avg'], 'frac %.3f' % r['frac'], 'algbytes %.1f MB' % (r['algorithmic_bytes_per_launch']/1e6), 'bad', d['config']['items_with_nonzero_status'], d['config']['non_finite_rewards'])
"; }
VARIANTS=("$@")
if [ ${#VARIANTS[@]} -eq 0 ]; then VARIANTS=("old=IPP_PATCH=0" "patch_w2=IPP_PATCH_WAVES=2" "patch_w1=IPP_PATCH_WAVES=1" "patch_w4=IPP_PATCH_WAVES=4"); fi
for rep in 1 2; do
  for v in "${VARIANTS[@]}"; do
    name=${v%%=*}; rest=${v#*=}
    IFS=',' read -ra kv <<< "$rest"
    run $name "${kv[@]}"
  done
done 2>&1 | tee -a $O/ab_patch.txt
