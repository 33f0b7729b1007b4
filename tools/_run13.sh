#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/proxy_bench.py 8 0 10
python3 tools/proxy_bench.py 8 3 10
rm -rf gpurun_out/prof_px
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_px -o t -- python3 tools/proxy_bench.py 8 0 5 > gpurun_out/px.json 2>/dev/null
cat gpurun_out/px.json
win=$(python3 -c "import json; print(5 * json.load(open('gpurun_out/px.json'))['ms_per_step'] + 0.2)")
python3 tools/rocpd_summary.py gpurun_out/prof_px/t_results.db $win | head -42
python3 tools/rocpd_summary.py gpurun_out/prof_px/t_results.db $win --gaps | head
rm -rf gpurun_out/prof_px
