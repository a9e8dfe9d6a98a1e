OUT=gpurun_out/tl4.txt
: > $OUT
IPP_TIMELINE_FILE=/tmp/tl.bin python tools/ab_kernels.py --window-rows 10 --order desc --rounds 30 t=$PWD/tools/probes/libipp_timing.so 2>&1 | grep -v amdgpu | tail -1 >> $OUT
python tools/timeline.py /tmp/tl.bin 4096 20 2>&1 | head -24 >> $OUT
cat $OUT
