run() { python $3bench.py --steps 40 --warmup 4 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'T=%d'%d['config']['tile_threads'], 'steps/s=%.0f'%d['value'], 'ms=%.3f'%d['ms_per_step'], 'GB/s=%.0f'%r['achieved'], 'gain_ms=%.3f'%r['kernel_ms_avg'], 'prep_ms=%.3f'%r['other_kernels_ms_avg']['k_prepare'], 'rank=%.1f'%d['config']['mean_rank_after_step'], 'bytes=%.3g/%.3g'%(r['algorithmic_bytes_per_launch'], r['full_column_formula_bytes_per_launch']))"; }
run wave "" ""
run fusedWG "--tile-threads 256" ""
run wave "" ""
run wave-w14 "--window-rows 14" ""
run wave-w100 "--window-rows 100" ""
