#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/run17_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run17_tests.log
tail -7 gpurun_out/run17_tests.log
python3 tools/proxy_bench.py 8 0 20
python bench.py --steps 10 --warmup 3 --cpu_queries 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print({k:r[k] for k in ('ms_per_step','value','ms_per_step_full_forward')}, r['roofline']['frac'])
print(r['shard_proxy_8']['proxy_ms'], r['shard_proxy_8']['projected_efficiency'], r['latency_config1']['ms_per_query'], r['latency_config1']['hip_graph']['ms_per_query'], r['config5']['ms_per_step'])"
