#!/bin/bash
# Round-end measurement on the GPU box: tests, bench lines, kernel trace, PMC passes. Outputs under gpurun_out/round/.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/round; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
python bench.py --no-cpu-baseline --window-rows 12 > $O/bench_w12.log 2>&1; tail -1 $O/bench_w12.log | cut -c1-300
python bench.py --no-cpu-baseline --window-rows 0 > $O/bench_exact.log 2>&1; tail -1 $O/bench_exact.log | cut -c1-300
python bench.py --no-cpu-baseline --tile-threads 64 > $O/bench_wave.log 2>&1; tail -1 $O/bench_wave.log | cut -c1-300
python bench.py --no-cpu-baseline --state dense --envs 256 > $O/bench_dense.log 2>&1; tail -1 $O/bench_dense.log | cut -c1-300
# the rows next to the hot path: candidate scoring, tree steps, other grid sizes, predict-only rate
{
  python tools/window_sweep.py --fixed-prior 10 12 0
  python tools/score_bench.py --steps 12; python tools/score_bench.py --steps 40 --window-rows 12; python tools/score_bench.py --state dense --steps 12
  python tools/score_bench.py --grid 100 --steps 12
  python tools/tree_bench.py; python tools/tree_bench.py --grid 200 --roots 1024 --root-steps 5
  for extra in "--predict-only" "--grid 100 --envs 32768 --episode-steps 16 --steps 20 --warmup 4" "--grid 200 --envs 1024 --episode-steps 10 --steps 10 --warmup 2 --predict-only"; do
    python bench.py --no-cpu-baseline $extra | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench $extra:', '%.4g env-steps/s' % d['value'], '%.4f ms/step' % d['ms_per_step'], r['kernel'], '%.4f ms' % r['kernel_ms_avg'], '%.0f GB/s' % r['achieved'])"
  done
} > $O/extras.txt 2>&1
grep -v amdgpu.ids $O/extras.txt | tail -18
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o kt -- python3 bench.py --no-cpu-baseline > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -2
python tools/trace_phases.py $O/trace/kt_kernel_trace.csv $O/trace.log > $O/trace_phases.txt 2>&1; cat $O/trace_phases.txt
bash tools/pmc_run.sh $O/pmc > $O/pmc_run.log 2>&1
python tools/pmc_summary.py $O/pmc 40 > $O/pmc_summary.json 2>$O/pmc_summary.err; head -c 120 $O/pmc_summary.json
# the other bench variants (exact factor mode, one-wave kernel, dense state): FETCH / WRITE passes only
for variant in "exact --window-rows 0" "wave --tile-threads 64" "dense --state dense --envs 256"; do
  set -- $variant; name=$1; shift
  PMC_SETS=traffic bash tools/pmc_run.sh $O/pmc_$name "$@" > $O/pmc_run_$name.log 2>&1
  python tools/pmc_summary.py $O/pmc_$name 40 > $O/pmc_summary_$name.json 2>>$O/pmc_summary.err
done
# keep the merge-back small: drop raw per-dispatch csvs except stats
find $O -path "*pmc*" -name "*.csv" -size +1M -delete; find $O/trace -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
