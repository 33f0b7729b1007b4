#!/bin/bash
cd $GRAFT_REPO_ROOT
( python tools/fuzz_parity.py 1500 5000 ego4d 0 ) > gpurun_out/soak2_r03.log 2>&1
grep -v amdgpu gpurun_out/soak2_r03.log | tail -4
