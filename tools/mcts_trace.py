"""Timeline of one device search from a rocprofv3 kernel trace: kernels and gaps of the LAST search of tools/mcts_readout.py.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python3 tools/mcts_readout.py 8 ; python tools/mcts_trace.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ipp::", "")[:60]))
rows.sort()
sel = [i for i, r in enumerate(rows) if r[2].startswith("k_mcts_select")]
n_waves = int(sys.argv[2]) if len(sys.argv) > 2 else 32
first = sel[-n_waves]
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_mcts_backup"))
seg = rows[first:last + 1]
span = seg[-1][1] - seg[0][0]
busy = defaultdict(lambda: [0, 0])
gap_after = defaultdict(lambda: [0, 0])
for a, b in zip(seg, seg[1:]):
    g = b[0] - a[1]
    gap_after[a[2]][0] += max(g, 0); gap_after[a[2]][1] += 1
for s, e, n in seg:
    busy[n][0] += e - s; busy[n][1] += 1
print(f"last search: {len(seg)} kernels over {span / 1e6:.2f} ms, busy {sum(v[0] for v in busy.values()) / 1e6:.2f} ms")
for n, (t, c) in sorted(busy.items(), key=lambda kv: -kv[1][0]):
    g = gap_after[n]
    print(f"  {n:60s} calls {c:4d}  busy {t / 1e3:9.1f} us  avg {t / c / 1e3:7.1f}   idle behind it {g[0] / 1e3:9.1f} us (avg {g[0] / max(g[1], 1) / 1e3:6.1f})")
w = [i for i, r in enumerate(seg) if r[2].startswith("k_mcts_select")]
a, b = w[len(w) // 2], w[len(w) // 2 + 1]
t0 = seg[a][0]
print("one wave of simulations:")
for s, e, n in seg[a:b]:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  {n}")
