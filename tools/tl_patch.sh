#!/bin/bash
# Timeline of one k_step_patch launch (IPP_TIMELINE build) + the A/B of the waves-per-item variants, each under its own timeout.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab; mkdir -p $O
for w in ${TL_WAVES:-2}; do
  echo "== timeline, $w waves per item"
  IPP_PATCH_WAVES=$w IPP_TIMELINE_FILE=$O/tl_w$w.bin timeout 300 python tools/ab_kernels.py --window-rows 10 --rounds 20 --order desc t=tools/probes/libipp_timing.so 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 60 python tools/timeline.py $O/tl_w$w.bin 4096 10
done 2>&1 | tee $O/timeline.txt
rm -f $O/*.bin
