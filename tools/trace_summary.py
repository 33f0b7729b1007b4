#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: time per kernel, GPU-busy vs wall for the last N ms."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
t_end = ev[-1][1]
window_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else (t_end - ev[0][0])
sel = [e for e in ev if e[0] >= t_end - window_ns]
busy = 0
cur_s, cur_e = sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = sel[-1][1] - sel[0][0]
print(f"window {wall / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms ({100 * busy / wall:.1f}%), {len(sel)} launches")
per = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    k = n.split("(")[0][-60:]
    per[k][0] += e - s
    per[k][1] += 1
for k, (ns, c) in sorted(per.items(), key=lambda x: -x[1][0])[:22]:
    print(f"{ns / 1e6:9.3f} ms {c:6d}  {k}")
