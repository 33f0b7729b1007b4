#!/usr/bin/env python3
"""The launch sequence of ONE single-query step (BASELINE configs[0]) out of a rocprofv3 rocpd database:
    rocprofv3 --kernel-trace -d out -o t -- python3 tools/latency_bench.py 1x1
    python3 tools/latency_trace.py out/t_results.db
prints the kernels of the last step (--full: from stage A's first launch) in launch order with start offsets, durations and the idle gap ahead of each."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ev = sorted(db.execute("select start, end, name from kernels").fetchall())
# the last step = back from the end to the previous launch of the step's first kernel (l2norm of the clip band)
names = [e[2].split("(")[0].replace("void ", "").replace("cone::", "") for e in ev]
first = names[-1]
i = len(ev) - 1
starts = [j for j, n in enumerate(names) if "l2norm_kernel" in n]
# a step has three l2norm launches near its head (clips, cls, tokens): take the last group
j = starts[-1]
while j - 1 in starts or (j - 2 in starts):
    j -= 1 if j - 1 in starts else 2
if "--full" in sys.argv:        # the WHOLE step: stage A too (its head is the clip band's l2norm, two l2norm launches earlier)
    j = starts[-3]
sel = list(zip(ev[j:], names[j:]))
t0 = sel[0][0][0]
prev_end = t0
busy = 0
for (s, e, _), n in sel:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n[:90]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"# {len(sel)} launches, span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
