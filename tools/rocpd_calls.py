#!/usr/bin/env python3
"""List the launches of one step from a rocprofv3 rocpd database in time order: offset, duration, grid, kernel.
usage: rocpd_calls.py results.db <first kernel of a step (substring)> [step index from the end, default 1]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
q = "select start, end, name" + (f", {gx}, {wx}" if gx and wx else "") + " from kernels order by start"
ev = db.execute(q).fetchall()
marks = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
i0 = marks[-back - 1]
i1 = marks[-back]
t0 = ev[i0][0]
prev_end = t0
for e in ev[i0:i1]:
    grid = f"{e[3] // max(e[4], 1):6d} x {e[4]:4d}" if len(e) > 3 else ""
    print(f"{(e[0] - t0) / 1e3:9.1f} us  +{(e[0] - prev_end) / 1e3:6.1f}  {(e[1] - e[0]) / 1e3:8.1f} us  {grid}  {e[2].split('(')[0][:70]}")
    prev_end = max(prev_end, e[1])
print(f"# step: {(ev[i1][0] - t0) / 1e3:.1f} us, {i1 - i0} launches; columns: {cols}")
