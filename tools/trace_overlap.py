#!/usr/bin/env python3
"""Timeline of the step kernel's dispatches from a rocprofv3 --kernel-trace CSV: per queue the gap between consecutive dispatches, the
overlap between queues, and the time during which at least one dispatch runs (union) per launch.
usage: python tools/trace_overlap.py <kernel_trace.csv> [kernel substring] [launches of the tail to analyse]"""
import csv, sys, collections
path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_step_patch"
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
rows = rows[-tail:]
t0 = rows[0][0]
by_q = collections.defaultdict(list)
for s, e, q, g in rows:
    by_q[q].append((s, e, g))
print(f"{len(rows)} dispatches of {name} on {len(by_q)} queues; first 12:")
for s, e, q, g in rows[:12]:
    print(f"  queue {q:>3}  grid {g:>7}  start {1e-3 * (s - t0):9.1f} us  end {1e-3 * (e - t0):9.1f}  dur {1e-3 * (e - s):7.1f}")
for q, lst in by_q.items():
    gaps = [1e-3 * (lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
    durs = [1e-3 * (e - s) for s, e, _ in lst]
    gaps.sort()
    print(f"queue {q}: {len(lst)} dispatches, duration mean {sum(durs) / len(durs):.1f} us, gap to the next dispatch of the queue: median {gaps[len(gaps) // 2]:.1f} mean {sum(gaps) / len(gaps):.1f} p90 {gaps[int(0.9 * len(gaps))]:.1f} us")
busy, end = 0, -1
for s, e, _, _ in rows:
    if s > end:
        busy += e - s; end = e
    elif e > end:
        busy += e - end; end = e
span = rows[-1][1] - rows[0][0]
print(f"union busy {1e-3 * busy:.1f} us of span {1e-3 * span:.1f} us ({busy / span:.3f}); per dispatch: busy {1e-3 * busy / len(rows):.2f} us, span {1e-3 * span / len(rows):.2f} us")
