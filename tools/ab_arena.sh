#!/bin/bash
# A/B of the arena's origin on the driver's protocol, fresh processes: torch tensor against VMM arenas (chunk = alignment)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/arena; mkdir -p $O
run() {  # tag, env assignments, bench args
    tag=$1; shift; envs=$1; shift
    for i in 1 2 3; do
        env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); c = r['config']
        print('$tag run $i', r['value'] / 1e6, 'M', r['ms_per_step'], 'ms  frac', r['roofline']['frac'], 'regions', c.get('region_ms_min'), c.get('region_ms_max'))
"
    done
}
{
for mode in "IPP_ARENA=torch" "IPP_ARENA=auto"; do
    run "cfg3share [$mode]" "$mode" --envs 32768 --grid 50
done
for mode in "IPP_ARENA=torch" "IPP_ARENA=auto"; do
    run "headline [$mode]" "$mode"
    run "cfg2 [$mode]" "$mode" --envs 32768 --grid 100 --episode-steps 16
done
} 2>&1 | tee $O/ab_arena.txt
