#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_pf
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pf -o t -- python3 tools/prefilter_bench.py --queries 64 --steps 5 > /dev/null 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_pf/t_results.db 21 | head -12
python3 - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/prof_pf/t_results.db")
ev = sorted(db.execute("select start, end, name from kernels").fetchall())[-12:]
t0 = ev[0][0]
for s, e, n in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {n.split('(')[0][:60]}")
PY
