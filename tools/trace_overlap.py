#!/usr/bin/env python3
"""Timeline of the step kernel's dispatches from a rocprofv3 --kernel-trace CSV.  A bench.py run holds stretches of the partitioned
schedule (one dispatch per group of envs, grid < the batch) and stretches with the whole batch as one dispatch; per stretch: the
dispatches' mean duration, and per STEP the wall time and the time during which at least one dispatch runs (union) -- the kernel time
of a step when its launches overlap.  Then, for the last `tail` dispatches, the per-queue gaps.
usage: python tools/trace_overlap.py <kernel_trace.csv> [kernel substring] [tail] [bench log holding the JSON line of the same run]"""
import collections
import csv
import json
import sys

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_step_patch"
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
full = max(g for _, _, _, g in rows)


def union(seg):
    busy, end = 0, -1
    for s, e, _, _ in seg:
        if s > end:
            busy += e - s; end = e
        elif e > end:
            busy += e - end; end = e
    return busy


print(f"{len(rows)} dispatches of {name}; whole batch = grid {full}")
print("stretches of >= 40 dispatches (in time order):")
seg, segs = [], []
for r in rows:
    cls = r[3] == full
    if seg and (seg[-1][3] == full) != cls:
        segs.append(seg); seg = []
    seg.append(r)
segs.append(seg)
for seg in segs:
    if len(seg) < 40:
        continue
    is_full = seg[0][3] == full
    queues = sorted({q for _, _, q, _ in seg})
    items = sum(g for _, _, _, g in seg)
    steps = items / full  # every step dispatches the whole batch once, in one piece or in groups
    seg_in = seg[len(queues):-len(queues)] if len(seg) > 4 * len(queues) else seg  # (without the ramp at either end)
    steps_in = sum(g for _, _, _, g in seg_in) / full
    span = max(e for _, e, _, _ in seg_in) - seg_in[0][0]
    durs = [1e-3 * (e - s) for s, e, _, _ in seg]
    print(f"  {'one dispatch per step' if is_full else 'groups on %d queues' % len(queues):24s} {len(seg):5d} dispatches = {steps:7.1f} steps: "
          f"dispatch duration mean {sum(durs) / len(durs):6.1f} us;  per step: wall {1e-3 * span / steps_in:6.1f} us, "
          f"at least one dispatch running {1e-3 * union(seg_in) / steps_in:6.1f} us")
rows = rows[-tail:]
t0 = rows[0][0]
by_q = collections.defaultdict(list)
for s, e, q, g in rows:
    by_q[q].append((s, e, g))
print(f"last {len(rows)} dispatches, on {len(by_q)} queues; first 12 of them:")
for s, e, q, g in rows[:12]:
    print(f"  queue {q:>3}  grid {g:>7}  start {1e-3 * (s - t0):9.1f} us  end {1e-3 * (e - t0):9.1f}  dur {1e-3 * (e - s):7.1f}")
for q, lst in by_q.items():
    gaps = sorted(1e-3 * (lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
    durs = [1e-3 * (e - s) for s, e, _ in lst]
    if gaps:
        print(f"queue {q}: {len(lst)} dispatches, duration mean {sum(durs) / len(durs):.1f} us, gap to the next dispatch of the queue: "
              f"median {gaps[len(gaps) // 2]:.1f} mean {sum(gaps) / len(gaps):.1f} p90 {gaps[int(0.9 * len(gaps))]:.1f} us")
busy = union(rows)
span = rows[-1][1] - rows[0][0]
print(f"union busy {1e-3 * busy:.1f} us of span {1e-3 * span:.1f} us ({busy / span:.3f}); per dispatch: busy {1e-3 * busy / len(rows):.2f} us, span {1e-3 * span / len(rows):.2f} us")
if len(sys.argv) > 4:
    d = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"bench.py of the same run: ms_per_step {d['ms_per_step']:.4f}; roofline.kernel_ms_avg {r['kernel_ms_avg']:.4f} ms per step "
          f"({r.get('launches_per_step', 1)} launches per step; the LAST stretch above is the leg it was measured on, with HIP events around every "
          f"dispatch), part_launch_ms_avg {r.get('part_launch_ms_avg') or 0:.4f}, single_launch_ms_avg {r.get('single_launch_ms_avg') or 0:.4f} "
          f"(the stretch of one dispatch per step before it)")
