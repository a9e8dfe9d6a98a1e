#!/bin/bash
# Copy the summaries of the last tools/round_check.sh run (gpurun_out/round) into profiles/<tag>_* (tracked).  usage: bash tools/save_profiles.sh r03
tag=${1:-r03}; R=gpurun_out/round; P=profiles
tail -1 $R/bench_default.log > $P/${tag}_bench_line.json
tail -1 $R/bench_driver.log > $P/${tag}_bench_line_driver.json
[ -f $R/split_pmc.json ] && cp $R/split_pmc.json $P/${tag}_split_pmc.json
cp $R/trace/kt_kernel_stats.csv $P/${tag}_bench_kernel_stats.csv 2>/dev/null || cp $(find $R/trace -name "*kernel_stats.csv" | head -1) $P/${tag}_bench_kernel_stats.csv
cp $(find $R/trace_cfg2 -name "*kernel_stats.csv" | head -1) $P/${tag}_cfg2_kernel_stats.csv
cp $(find $R/trace_tree -name "*kernel_stats.csv" | head -1) $P/${tag}_tree_kernel_stats.csv
cp $(find $R/trace_mcts -name "*kernel_stats.csv" | head -1) $P/${tag}_mcts_kernel_stats.csv
[ -f $R/trace_phases.txt ] && cp $R/trace_phases.txt $P/${tag}_trace_phases.txt
cp $R/pmc_summary.json $P/${tag}_pmc_summary.json
cp $R/pmc_summary_w12.json $P/${tag}_pmc_summary_w12.json
cp $R/pmc_summary_cfg2.json $P/${tag}_pmc_summary_cfg2.json
cp $R/patch_probe.txt $P/${tag}_patch_probe.txt
cp $R/timeline.txt $P/${tag}_timeline.txt
cp $R/write_probe_calibration.txt $P/${tag}_write_probe_calibration.txt
grep -v amdgpu.ids $R/extras.txt > $P/${tag}_extras.txt
tail -3 $R/pytest_gpu.log > $P/${tag}_pytest_gpu.txt
for f in ab_arena issue_probe prio_probe ab_parts trace_overlap parts_probe grf_bench region_sweep ab_split timeline_split mcts_trace; do [ -f $R/$f.txt ] && grep -v amdgpu.ids $R/$f.txt > $P/${tag}_$f.txt; done
ls -la $P | grep ${tag}_
