#!/bin/bash
# Collect PMC counters for the bench kernels (separate passes; no trace domains combined with --pmc).
# Every pass runs under `timeout`: a pass whose counter set is rejected can hang in the profiler's teardown.
# usage (on the GPU box): [PMC_SETS=traffic] bash tools/pmc_run.sh <outdir> [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
python3 bench.py --print-args "$@" > "$out/bench_args.json"
if [ -n "$PMC_CUSTOM" ]; then  # ';'-separated counter sets (diagnostics)
  IFS=';' read -ra sets <<< "$PMC_CUSTOM"
elif [ "$PMC_SETS" = "traffic" ]; then  # HBM bytes only (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  sets=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum")
else
  sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
        "FETCH_SIZE"
        "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
        "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
        "GRBM_GUI_ACTIVE")
fi
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -o pmc -- python3 bench.py --no-cpu-baseline --no-extra "$@" > "$out/pass$i.log" 2>&1
  tail -1 "$out/pass$i.log" | cut -c1-200
done
find "$out" -name "*.csv" | head -20
