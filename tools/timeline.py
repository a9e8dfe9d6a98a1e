#!/usr/bin/env python3
"""
Occupancy timeline of the last fused step launch of a -DIPP_PHASE_TIMING=1 build (tools/probes/libipp_timing.so):
reads the dump made through IPP_TIMELINE_FILE (per item: workgroup start, end of phase A, last wave's exit; 100 MHz
wall clock) and prints how many workgroups were resident / streaming over time.

usage: python tools/timeline.py dump.bin [n_items] [bucket_us]
"""
import sys

import numpy as np


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    bucket = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)[:n].astype(np.float64)
    t0 = t[:, 0].min()
    start, mid, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0  # us
    ok = t[:, 2] > 0
    print(f"items {n} (with marks: {int(ok.sum())}), kernel span {end[ok].max():.1f} us")
    print(f"phase A  mean {np.mean(mid[ok] - start[ok]):.1f} us   p10 {np.percentile(mid[ok] - start[ok], 10):.1f}  p90 {np.percentile(mid[ok] - start[ok], 90):.1f}")
    print(f"phase B  mean {np.mean(end[ok] - mid[ok]):.1f} us   p10 {np.percentile(end[ok] - mid[ok], 10):.1f}  p90 {np.percentile(end[ok] - mid[ok], 90):.1f}  max {np.max(end[ok] - mid[ok]):.1f}")
    if np.all(t[ok, 3] > 0):  # sub-phases of phase A (marks in prepare_item_ex)
        names = ["inputs + header", "ground truth / mean + observation", "gather of U[F,:]", "tables, mask, -HT rows"]
        cuts = [t[:, 0], t[:, 3], t[:, 4], t[:, 5], t[:, 1]]
        for nm, a, b in zip(names, cuts[:-1], cuts[1:]):
            d = (b[ok] - a[ok]) / 100.0
            print(f"  {nm:36s} mean {d.mean():5.1f} us   p10 {np.percentile(d, 10):5.1f}  p90 {np.percentile(d, 90):5.1f}")
    if np.all(t[ok, 6] > 0):
        first_out, solved = (t[:, 6] - t0) / 100.0, (t[:, 7] - t0) / 100.0
        print(f"wave 0's m x m algebra after phase A: mean {np.mean(solved[ok] - mid[ok]):.1f} us   p90 {np.percentile(solved[ok] - mid[ok], 90):.1f}")
        print(f"first wave out -> last wave out: mean {np.mean(end[ok] - first_out[ok]):.1f} us   p90 {np.percentile(end[ok] - first_out[ok], 90):.1f}")
    # slot refill: for every workgroup start after the first wave of starts, the time since the latest earlier end
    order_e = np.sort(end[ok])
    late = np.sort(start[ok])[min(1024, int(ok.sum()) - 1):]
    if len(late):
        gaps = []
        for k, s0 in enumerate(late):  # k-th late start reuses (at best) the slot of the k-th end
            gaps.append(s0 - order_e[min(k, len(order_e) - 1)])
        print(f"k-th late start minus k-th end (slot refill delay): mean {np.mean(gaps):.1f} us   p10 {np.percentile(gaps, 10):.1f}  p90 {np.percentile(gaps, 90):.1f}")
    import os
    if os.path.exists(path + ".units"):
        ut = np.fromfile(path + ".units", dtype=np.uint64).reshape(-1, 2, 8, 4)
        for i in np.argsort(-end)[:3]:
            print(f" units of item {i} (start {start[i]:.1f}, end {end[i]:.1f} us):")
            for w in range(2):
                for k in range(8):
                    a = ut[i, w, k]
                    if a[1] == 0:
                        continue
                    ts = [(float(x) - t0) / 100.0 for x in a[1:]]
                    print(f"   wave {w} unit {int(a[0] >> np.uint64(32))} rows {int(a[0] & np.uint64(0xffffffff)):4d}  start {ts[0]:6.1f}  stream done {ts[1]:6.1f} (+{ts[1] - ts[0]:5.1f})  end {ts[2]:6.1f} (+{ts[2] - ts[1]:4.1f})")
    print(f"last workgroup start at {start[ok].max():.1f} us")
    # the items that end last: where did their time go
    last = np.argsort(-end)[:12]
    print(" item   start  phaseA(header gather tables)  algebra_done  first_wave_out  end   [us]")
    for i in last:
        sub = [(t[i, k] - t0) / 100.0 for k in (3, 4, 5)] if t[i, 3] > 0 else [0, 0, 0]
        print(f" {i:5d} {start[i]:7.1f} {mid[i] - start[i]:6.1f} ({sub[0] - start[i]:5.1f} {sub[2] - sub[0]:5.1f} {mid[i] - sub[2]:5.1f}) "
              f"{(t[i, 7] - t0) / 100.0 - mid[i]:10.1f} {(t[i, 6] - t0) / 100.0:12.1f} {end[i]:7.1f}")
    edges = np.arange(0.0, end[ok].max() + bucket, bucket)
    print(" window[us]  resident  in phase A  streaming")
    for a, b in zip(edges[:-1], edges[1:]):
        c = 0.5 * (a + b)
        res = np.sum((start <= c) & (end > c) & ok)
        pa = np.sum((start <= c) & (mid > c) & ok)
        print(f" {a:6.0f}-{b:<6.0f} {res:8d} {pa:10d} {res - pa:10d}")


if __name__ == "__main__":
    main()
