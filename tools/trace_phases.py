#!/usr/bin/env python3
"""
Per-phase mean duration of the bench's headline kernel from a rocprofv3 kernel trace of `python bench.py`:
the *_kernel_stats.csv average runs over every launch of the process (40 pre-roll steps with growing ranks, warm-up,
timed region, roofline leg); only the last block is what bench.py's roofline.kernel_ms_avg measures (HIP events
attached to the same dispatches), so that is the number the two must agree on.

usage: python tools/trace_phases.py <kt_kernel_trace.csv> <bench json line file> [episode_steps]
"""
import csv
import json
import sys


def main():
    trace, line_file = sys.argv[1], sys.argv[2]
    pre = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    line = [l for l in open(line_file) if l.startswith("{")][-1]
    d = json.loads(line)
    name, steps, warm = d["roofline"]["kernel"], d["steps"], d["warmup"]
    rows = [r for r in csv.DictReader(open(trace)) if ("ipp::" + name) in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    cuts = [("pre-roll", pre), ("warm-up", warm), ("timed region", steps), ("roofline leg", steps)]
    print(f"{name}: {len(dur)} launches, mean over all {sum(dur) / len(dur):.4f} ms")
    pos = 0
    for label, n in cuts:
        part = dur[pos:pos + n]
        if part:
            print(f"  {label:13s} launches {pos:3d}..{pos + len(part) - 1:3d}: mean {sum(part) / len(part):.4f} ms")
        pos += n
    print(f"bench.py of the same run: roofline.kernel_ms_avg {d['roofline']['kernel_ms_avg']:.4f} ms, ms_per_step {d['ms_per_step']:.4f}")


if __name__ == "__main__":
    main()
