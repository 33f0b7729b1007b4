#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_two_ranks or distributed_drivers or virtual_rank" > gpurun_out/run4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run4_tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/run4_bench.json 2> gpurun_out/run4_bench.err
tail -12 gpurun_out/run4_tests.log; tail -3 gpurun_out/run4_bench.err; cat gpurun_out/run4_bench.json
