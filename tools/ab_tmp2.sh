OUT=gpurun_out/ht.txt
: > $OUT
timeout 900 python -m pytest tests/test_hip_rect_meta.py tests/test_hip_parity.py tests/test_hip_window.py tests/test_hip_edge_cases.py tests/test_hip_sharding.py tests/test_hip_autoreset.py -q 2>&1 | tr -cd "[:print:]\n" | tail -3 >> $OUT
A=$PWD/ipp-rl_amd/lib/libipp_hip.so
B=$PWD/tools/probes/libipp_prev.so
python tools/ab_kernels.py --window-rows 10 --order desc,desc --rounds 90 new=$A prev=$B 2>&1 | grep -v amdgpu | tail -2 >> $OUT
python tools/ab_kernels.py --window-rows 10 --order desc,desc --rounds 90 prev=$B new=$A 2>&1 | grep -v amdgpu | tail -2 >> $OUT
python tools/ab_kernels.py --window-rows 12 --order desc,desc --rounds 90 prev=$B new=$A 2>&1 | grep -v amdgpu | tail -2 >> $OUT
line() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms kernel', r['kernel'], round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],3))" >> $OUT 2>&1; }
python bench.py --no-extra --no-cpu-baseline | line "cfg1 new"
IPP_HIP_LIB=$B python bench.py --no-extra --no-cpu-baseline | line "cfg1 prev"
python bench.py --no-extra --no-cpu-baseline | line "cfg1 new"
IPP_HIP_LIB=$B python bench.py --no-extra --no-cpu-baseline | line "cfg1 prev"
cat $OUT
