run() { python $3bench.py --steps 40 --warmup 4 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'T=%d'%d['config']['tile_threads'], 'steps/s=%.0f'%d['value'], 'ms=%.3f'%d['ms_per_step'], 'GB/s=%.0f'%r['achieved'], 'gain_ms=%.3f'%r['kernel_ms_avg'], 'prep_ms=%.3f'%r['other_kernels_ms_avg']['k_prepare'], 'rank=%.1f'%d['config']['mean_rank_after_step'])"; }
run base "" ""
for q in 64 128; do IPP_QCHUNK=$q run base-q$q "" ""; done
for w in 5 6; do for q in 64 128 1024; do IPP_QCHUNK=$q IPP_HIP_LIB=$PWD/ipp-rl_amd/lib/ab/libipp_w$w.so run w$w-q$q "" ""; done; done
run base "" ""
