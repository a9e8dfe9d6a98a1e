#!/bin/bash
# A/B of engine builds (tools/variant.sh) on the headline workload, one box: usage: [BENCH_ARGS=..] bash tools/ab_libs.sh name[=ENV=VAL,..] ...   (name "main" = the in-tree library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    name=${v%%=*}; rest=""; [[ "$v" == *=* ]] && rest=${v#*=}
    lib=""; [ "$name" != "main" ] && lib="IPP_HIP_LIB=$PWD/tools/probes/libipp_$name.so IPP_AB_OLD_LIB=1"
    IFS=',' read -ra kv <<< "$rest"
    env $lib "${kv[@]}" python bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$v', 'value %.2f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms_avg'], 'frac %.3f' % r['frac'], 'alg MB %.1f' % (r['algorithmic_bytes_per_launch']/1e6), 'bad', c['items_with_nonzero_status'], c['non_finite_rewards'])"
  done
done 2>&1 | tee -a $O/ab_libs.txt
