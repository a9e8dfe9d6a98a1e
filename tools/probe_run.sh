#!/bin/bash
# patch_probe: GB/s per address pattern, then FETCH_SIZE per pattern (separate counter pass), then the TLB / cache counters
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/probe; mkdir -p $O
timeout 120 ./tools/probes/patch_probe 4096 32 117 2>&1 | tee $O/patch_probe.txt
timeout 120 ./tools/probes/patch_probe 4096 48 351 2>&1 | tee -a $O/patch_probe.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o pp -- ./tools/probes/patch_probe 4096 32 117 > $O/fetch.log 2>&1
python3 - <<'PY' | tee -a gpurun_out/probe/patch_probe.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("gpurun_out/probe/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"FETCH_SIZE {k:30s} mean over {len(v)} dispatches: {sum(v)/len(v):12.0f} KiB")
PY
timeout 120 rocprofv3 --list-avail 2>/dev/null | grep -i -o -E "(TCP|TCC|TA|UTCL|GL2|TLB)[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
find $O -name "*.csv" -size +1M -delete
