#!/usr/bin/env python3
"""Plain timeline of a rocprofv3 --kernel-trace CSV: the last `n` dispatches whose name holds one of the substrings, per line
start / end (us from the first listed), duration, queue, grid, short name.  usage: trace_dump.py <kernel_trace.csv> [n] [substr,substr..]"""
import csv
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
subs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["k_step", "k_grf"]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        if any(s in r["Kernel_Name"] for s in subs):
            nm = r["Kernel_Name"].split("(")[0].replace("void ipp::", "").replace("ipp::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0) or 0), nm))
rows.sort()
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = rows[-(n + skip):len(rows) - skip] if skip else rows[-n:]
t0 = rows[0][0]
for s, e, q, g, nm in rows:
    print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - t0):9.1f}  {1e-3 * (e - s):7.1f} us  q{q:>3s} grid {g:7d}  {nm[:60]}")
