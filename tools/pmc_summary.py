#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_run.sh output): per-kernel mean of every counter over dispatches."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # ignore the first `skip` dispatches of each kernel (pre-roll)
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = defaultdict(int)
    with open(path) as fh:
        rows = list(csv.DictReader(fh))
    by_dispatch = defaultdict(dict)
    for r in rows:
        by_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (did, name), counters in sorted(by_dispatch.items(), key=lambda kv: int(kv[0][0])):
        short = name.split("(")[0].replace("void ", "")
        seen[short] += 1
        if seen[short] <= skip:
            continue
        for c, v in counters.items():
            acc[short][c].append(v)
out = {}
for k, cs in acc.items():
    if not k.startswith("ipp::"):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
key_path = os.path.join(root, "bench_args.json")  # written by tools/pmc_run.sh: the bench command line the passes ran
if os.path.exists(key_path):
    with open(key_path) as fh:
        out["_bench_args"] = json.load(fh)
print(json.dumps(out, indent=1, sort_keys=True))
