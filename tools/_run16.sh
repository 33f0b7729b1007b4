#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "distributed or virtual_rank or bench_two" > gpurun_out/run16_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run16_tests.log
tail -4 gpurun_out/run16_tests.log
python3 tools/proxy_bench.py 8 0 20
python3 tools/proxy_bench.py 8 5 20
python3 tools/proxy_bench.py 4 1 20
python3 tools/proxy_bench.py 2 1 10
