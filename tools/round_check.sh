#!/bin/bash
# Round-end measurement on the GPU box: tests, the bench line, kernel trace, PMC passes, write-counter calibration.
# Outputs under gpurun_out/round/ (copy the summaries to profiles/rNN_*).  usage: bash tools/round_check.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/round; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
# the DRIVER's protocol first (20-step regions): this is the line BENCH_rNN.json records -- keep it as profiles/rNN_bench_line_driver.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -1 $O/bench_driver.log | cut -c1-300
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
# region length against the per-step time (fill / drain of the two-group pipeline, the staged blocks inside short regions)
for k in 10 20 50 200; do python bench.py --no-cpu-baseline --no-extra --steps $k --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('steps', d['steps'], 'value %.2f M' % (d['value']/1e6), 'ms/step %.4f' % d['ms_per_step'], 'host issue %.4f' % (c.get('host_issue_ms_per_step') or 0), 'regions', ['%.4f' % x for x in c['region_ms_per_step']])"; done > $O/region_sweep.txt 2>&1; cat $O/region_sweep.txt
# the split step (prologue kernel + unit kernel) against the fused kernel (round 5's question; ROUND_SPLIT=1 repeats it)
if [ "$ROUND_SPLIT" = "1" ]; then
# the split step (prologue kernel + unit kernel) against the fused kernel, same box: headline, one launch per step, configs[2], configs[3] share
{ CFG="--parts 2" bash tools/ab_cfg.sh "IPP_SPLIT=0" "IPP_SPLIT=1" "IPP_SPLIT=1 IPP_SPLIT_WAVES=1"
  CFG="--parts 1 --steps 100" bash tools/ab_cfg.sh "IPP_SPLIT=0" "IPP_SPLIT=1" "IPP_SPLIT=1 IPP_SPLIT_WAVES=1"
  CFG="--grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4" bash tools/ab_cfg.sh "IPP_SPLIT=0" "IPP_SPLIT=1 IPP_SPLIT_WAVES=1"
  CFG="--grid 50 --envs 32768 --steps 20 --warmup 4" bash tools/ab_cfg.sh "IPP_SPLIT=0" "IPP_SPLIT=1 IPP_SPLIT_WAVES=1"; } > $O/ab_split.txt 2>&1
[ -f tools/probes/libipp_timing.so ] && { for w in 1 3; do echo "=== split step, prologue kernel with $w waves per item, 2 groups"; IPP_SPLIT_WAVES=$w python tools/timeline_split.py 2 2>&1 | grep -v amdgpu.ids | head -40; done; } > $O/timeline_split.txt 2>&1
fi
# round 6: the arena's origin (torch tensor against the virtual-memory API), fresh processes on the driver's protocol
bash tools/ab_arena.sh > /dev/null 2>&1; cp gpurun_out/arena/ab_arena.txt $O/ab_arena.txt
{
  python tools/compat_bench.py
  python tools/score_bench.py --steps 12
  python tools/score_bench.py --grid 100 --steps 12
} > $O/extras.txt 2>&1
grep -v amdgpu.ids $O/extras.txt | tail -12
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# kernel trace of the headline command line (no extras: one workload per trace)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o kt -- python3 bench.py --no-cpu-baseline --no-extra > $O/trace.log 2>&1
python tools/trace_overlap.py $O/trace/kt_kernel_trace.csv k_step_patch 300 $O/trace.log > $O/trace_overlap.txt 2>&1; grep -v "^  queue" $O/trace_overlap.txt | head -12
# and of configs[2]
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2 -o kt -- python3 bench.py --no-cpu-baseline --no-extra --grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4 > $O/trace_cfg2.log 2>&1
# configs[4]: the tree wave (k_tree_patch) and the device-side search (k_mcts_*)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tree -o kt -- python3 tools/tree_wave.py --reps 2 > $O/trace_tree.log 2>&1; tail -1 $O/trace_tree.log | cut -c1-300
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mcts -o kt -- python3 tools/mcts_bench.py --reps 2 > $O/trace_mcts.log 2>&1; tail -1 $O/trace_mcts.log | cut -c1-300
# one search of 1024 roots x 256 simulations, 8 in flight: host times, then kernels and gaps of the last search from a kernel trace
{ timeout 300 python tools/mcts_readout.py 8 2>&1 | grep -v amdgpu.ids | head -2
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_search -o kt -- python3 tools/mcts_readout.py 8 > /dev/null 2>&1
  python tools/mcts_trace.py $O/trace_search; } > $O/mcts_trace.txt 2>&1; head -9 $O/mcts_trace.txt
# PMC passes (separate runs, counters only)
bash tools/pmc_run.sh $O/pmc --parts 1 > $O/pmc_run.log 2>&1
python tools/pmc_summary.py $O/pmc 40 > $O/pmc_summary.json 2>$O/pmc_summary.err; head -c 120 $O/pmc_summary.json
if [ "$ROUND_SPLIT" = "1" ]; then
# the split step's kernels (unit kernel: waves' wait share, traffic; prologue kernel) -- its summary is NOT named *pmc_summary*: bench.py
# replays traffic from those, and the split run has the same command line
IPP_SPLIT=1 PMC_CUSTOM="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES;FETCH_SIZE;WRITE_SIZE TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" bash tools/pmc_run.sh $O/pmc_split --parts 1 > $O/pmc_run_split.log 2>&1
python tools/pmc_summary.py $O/pmc_split 40 > $O/split_pmc.json 2>>$O/pmc_summary.err
fi
PMC_SETS=traffic bash tools/pmc_run.sh $O/pmc_w12 --parts 1 --shuffle-prior > $O/pmc_run_w12.log 2>&1
python tools/pmc_summary.py $O/pmc_w12 40 > $O/pmc_summary_w12.json 2>>$O/pmc_summary.err
bash tools/pmc_run.sh $O/pmc_cfg2 --parts 1 --grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4 > $O/pmc_run_cfg2.log 2>&1
python tools/pmc_summary.py $O/pmc_cfg2 16 > $O/pmc_summary_cfg2.json 2>>$O/pmc_summary.err
# address-pattern probe of the column layouts (GB/s + FETCH_SIZE per pattern: the FETCH_SIZE calibration on the step kernel's own
# request shapes) and the occupancy timeline of one k_step_patch launch (needs `make -C ipp-rl_amd/csrc timeline` before gpurun)
# round 4: issue cost of the kernels' instruction kinds, wave priority, the partitioned schedule against one launch per step
rm -f gpurun_out/ab/ab_parts.txt; bash tools/ab_parts.sh 1 2 3 > /dev/null 2>&1; cp gpurun_out/ab/ab_parts.txt $O/ab_parts.txt
python tools/parts_probe.py 2>&1 | grep -v amdgpu.ids > $O/parts_probe.txt
python tools/grf_bench.py 50:102 50:819 100:2048 > $O/grf_bench.txt 2>&1; IPP_GRF_FFT=0 python tools/grf_bench.py 50:102 50:819 100:2048 >> $O/grf_bench.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/patch_probe.hip -o tools/probes/patch_probe || echo "patch_probe build failed"
[ -x tools/probes/patch_probe ] && bash tools/probe_run.sh > $O/probe_run.log 2>&1 && cp gpurun_out/probe/patch_probe.txt $O/patch_probe.txt
[ -f tools/probes/libipp_timing.so ] && TL_WAVES=2 bash tools/tl_patch.sh > $O/tl_patch.log 2>&1 && cp gpurun_out/ab/timeline.txt $O/timeline.txt
# WRITE_SIZE / FETCH_SIZE calibration on known byte counts (tools/probes/write_probe.hip; the binary is git-ignored: build it here)
hipcc --offload-arch=gfx950 -O3 tools/probes/write_probe.hip -o tools/probes/write_probe || echo "write_probe build failed"
for c in WRITE_SIZE FETCH_SIZE; do
  [ -x tools/probes/write_probe ] || continue
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/wp_$c -o wp -- ./tools/probes/write_probe 1024 > $O/wp_$c.log 2>&1
done
python - <<'PY' > $O/write_probe_calibration.txt 2>&1
import csv, glob, collections
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/round/wp_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{c:10s} {k:16s} mean over {len(v)} dispatches: {sum(v)/len(v):12.0f} KiB for 1048576 KiB written once -> ratio {sum(v)/len(v)/1048576:.3f}")
PY
cat $O/wp_WRITE_SIZE.log | grep -v amdgpu; cat $O/write_probe_calibration.txt
find $O -path "*pmc*" -name "*.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -path "*wp_*" -name "*.csv" -size +1M -delete
du -sh $O
