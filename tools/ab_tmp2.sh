OUT=gpurun_out/opt.txt
: > $OUT
A=$PWD/ipp-rl_amd/lib/libipp_hip.so
python tools/ab_kernels.py --window-rows 10 --order desc,desc,desc --rounds 90 o3=$A o2=$PWD/tools/probes/libipp_o2.so os=$PWD/tools/probes/libipp_os.so 2>&1 | grep -v amdgpu | tail -3 >> $OUT
python tools/ab_kernels.py --window-rows 12 --order desc,desc,desc --rounds 90 os=$PWD/tools/probes/libipp_os.so o3=$A o2=$PWD/tools/probes/libipp_o2.so 2>&1 | grep -v amdgpu | tail -3 >> $OUT
cat $OUT
