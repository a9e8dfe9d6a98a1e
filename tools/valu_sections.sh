#!/bin/bash
# usage (GPU box): bash tools/valu_sections.sh   (needs tools/probes/libipp_exit.so: bash tools/variant.sh exit "-DIPP_EXIT_POINTS=1")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/valu; rm -rf $O; mkdir -p $O
export IPP_HIP_LIB=tools/probes/libipp_exit.so
for p in 0 1 2 3 4 5 6 7 8 9; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/p$p -o p -- python3 tools/valu_sections.py $p > $O/p$p.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/valu/valu_sections.txt
import csv, glob, collections
names = ["whole kernel", "1 header", "2 + rectangle tests, tables", "3 + gather, records", "4 + m x m algebra / observation", "5 + units"]
rows = {}
for p in range(10):
    f = glob.glob(f"gpurun_out/valu/p{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_step_patch" in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = per[max(per)]
    rows[p] = last
print("instructions of ONE launch (4096 items, configs[1] steady state), millions:  VALU   SALU   LDS   VMEM_RD")
names += ["whole kernel WITHOUT the units' prior term", "whole kernel WITHOUT the row stream", "whole kernel WITHOUT L^-1 in the epilogue", "whole kernel WITHOUT the units' stores"]
for p in [1, 2, 3, 4, 5, 0, 6, 7, 8, 9]:
    if p in rows:
        r = rows[p]
        print(f"  {'up to ' if p and p < 6 else ''}{names[p]:44s} {r.get('SQ_INSTS_VALU', 0) / 1e6:7.2f} {r.get('SQ_INSTS_SALU', 0) / 1e6:7.2f} {r.get('SQ_INSTS_LDS', 0) / 1e6:7.2f} {r.get('SQ_INSTS_VMEM_RD', 0) / 1e6:7.3f}")
PY
find $O -name "*.csv" -size +1M -delete
