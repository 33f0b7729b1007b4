#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "distributed or virtual_rank or bench_two or window_table or end_to_end or config5 or config2 or prefilter_batched or stage_a" > gpurun_out/run14_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run14_tests.log
tail -6 gpurun_out/run14_tests.log
python3 tools/proxy_bench.py 8 0 10
python3 tools/proxy_bench.py 8 5 10
python3 tools/proxy_bench.py 8 7 10
python3 tools/proxy_bench.py 2 1 10
