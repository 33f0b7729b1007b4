#!/usr/bin/env python3
"""Launch sequence of the LAST step out of a rocprofv3 rocpd database: kernels in launch order with start offsets, durations
and the idle gap ahead of each.  A step starts at the first l2norm_kernel launch of its head (clips, cls, tokens).
usage: step_trace.py results.db [min_gap_us_to_flag]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ev = sorted(db.execute("select start, end, name from kernels").fetchall())
names = [e[2].split("(")[0].replace("void ", "").replace("cone::", "") for e in ev]
starts = [j for j, n in enumerate(names) if "l2norm_kernel" in n]
# group l2norm launches that are close together (one step has three near its head); take the first of the last group
j = starts[-1]
while True:
    prev = [x for x in starts if x < j and j - x <= 12]
    if not prev:
        break
    j = prev[0]
sel = list(zip(ev[j:], names[j:]))
t0 = sel[0][0][0]
prev_end, busy, gaps = t0, 0, 0.0
for (s, e, _), n in sel:
    g = (s - prev_end) / 1e3
    gaps += max(g, 0.0)
    print(f"{(s - t0) / 1e3:9.1f} us  +{g:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n[:100]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"# {len(sel)} launches, span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {gaps:.1f} us")
