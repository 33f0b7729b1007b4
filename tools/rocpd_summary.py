#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (default output of ROCm 7.2's `rocprofv3 --kernel-trace --stats`):
per-kernel total / average / count over the last WINDOW ms of the trace, GPU-busy vs wall, as CSV on stdout.
usage: rocpd_summary.py results.db [window_ms] [--gaps]
--gaps: instead of the per-kernel table, list the GPU-idle gaps > 0.2 ms inside the window (start offset, length, the
kernels on either side) -- where the device waits for the host (between steps: D2H + list building)."""
import sqlite3
import sys
from collections import defaultdict

gaps = "--gaps" in sys.argv
if gaps:
    sys.argv.remove("--gaps")
db = sqlite3.connect(sys.argv[1])
ev = sorted(db.execute("select start, end, name from kernels").fetchall())
t_end = max(e[1] for e in ev)
window_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else t_end - ev[0][0]
sel = [e for e in ev if e[0] >= t_end - window_ns]
busy, (cs, ce) = 0, sel[0][:2]
for s, e, _ in sel[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
wall = max(e[1] for e in sel) - sel[0][0]
print(f"# window {wall / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms ({100 * busy / wall:.1f}%), {len(sel)} launches")
if gaps:
    print("offset_ms,gap_ms,after_kernel,before_kernel")
    end, last = sel[0][1], sel[0][2]
    tot = 0.0
    for s0, e0, n0 in sel[1:]:
        if s0 - end > 0.2e6:
            print(f"{(end - sel[0][0]) / 1e6:.3f},{(s0 - end) / 1e6:.3f},\"{last.split('(')[0]}\",\"{n0.split('(')[0]}\"")
            tot += (s0 - end) / 1e6
        if e0 > end:
            end, last = e0, n0
    print(f"# total of listed gaps {tot:.3f} ms")
    sys.exit(0)
per = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    per[n.split("(")[0]][0] += e - s
    per[n.split("(")[0]][1] += 1
print("kernel,calls,total_ms,avg_us,percent_of_busy")
for k, (ns, c) in sorted(per.items(), key=lambda x: -x[1][0]):
    print(f"\"{k}\",{c},{ns / 1e6:.3f},{ns / c / 1e3:.1f},{100 * ns / busy:.2f}")
