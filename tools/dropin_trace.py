#!/usr/bin/env python3
"""Launch sequence of the LAST drop-in batch out of a rocprofv3 rocpd database (tools/dropin_bench.py): kernels in launch
order with start offsets, durations and the idle gap ahead of each.  A batch starts at its first scan_lengths launch."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ev = sorted(db.execute("select start, end, name from kernels").fetchall())
names = [e[2].split("(")[0].replace("void ", "").replace("cone::", "") for e in ev]
marker = sys.argv[2] if len(sys.argv) > 2 else "scan_lengths_kernel"
starts = [j for j, n in enumerate(names) if marker in n]
# the padded entry scans twice (clips, tokens) + once in forward_packed: take the first of the last group of three
j = starts[-3] if len(starts) >= 3 and marker == "scan_lengths_kernel" else starts[-1]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sel = list(zip(ev[j - back:], names[j - back:]))
t0 = sel[0][0][0]
prev_end, busy = t0, 0
for (s, e, _), n in sel:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n[:100]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"# {len(sel)} launches, span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
