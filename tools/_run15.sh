#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/proxy_bench.py 8 0 10
rm -rf gpurun_out/prof_px
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_px -o t -- python3 tools/proxy_bench.py 8 0 5 > gpurun_out/px.json 2>/dev/null
cat gpurun_out/px.json
win=$(python3 -c "import json; print(5 * json.load(open('gpurun_out/px.json'))['ms_per_step'] + 0.2)")
python3 tools/rocpd_summary.py gpurun_out/prof_px/t_results.db $win | head -30
python3 tools/rocpd_summary.py gpurun_out/prof_px/t_results.db $win --gaps | head
python3 - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/prof_px/t_results.db")
ev = sorted(db.execute("select start, end, name from kernels").fetchall())
t_end = ev[-1][1]
sel = [e for e in ev if e[0] >= t_end - 9.0e6]
t0 = sel[0][0]
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  {n.split('(')[0][:70]}")
PY
rm -rf gpurun_out/prof_px
