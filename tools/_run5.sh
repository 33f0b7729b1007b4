#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out
rm -rf $out/prof_gap
rocprofv3 --kernel-trace --stats -d $out/prof_gap -o t -- python3 bench.py --steps 3 --warmup 1 --cpu_queries 0 --no_extras > $out/gap_step.json 2> $out/gap.log
win=$(python3 -c "import json,sys; print(3 * json.load(open('$out/gap_step.json'))['ms_per_step'] + 0.5)")
python3 tools/rocpd_summary.py $out/prof_gap/t_results.db $win --gaps
python3 tools/rocpd_summary.py $out/prof_gap/t_results.db $win | head -40
python3 tools/host_profile.py 2>&1 | tail -30
