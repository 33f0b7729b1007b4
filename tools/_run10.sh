#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prefilter or stage_a or topk or ctx_sharded or mad_scale or localizer or config5" > gpurun_out/run10_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/run10_tests.log
tail -5 gpurun_out/run10_tests.log
python3 tools/prefilter_bench.py --queries 1,64 --steps 8 2>&1 | cut -c 140-420
python3 - <<'PY'
import torch, sys, time
sys.path.insert(0,'.')
from cone_amd import ops
# big-k and odd shapes against torch.sort(stable)
g = torch.Generator().manual_seed(0)
for nq, n, k in ((3, 100001, 30), (2, 9000, 256), (2, 9000, 300), (5, 50000, 1), (1, 8193, 64), (4, 70000, 200), (2, 300000, 30)):
    x = torch.randint(0, 200, (nq, n), generator=g).float().cuda()
    idx, val = ops.topk_windows(x, k)
    sv, si = torch.sort(x, dim=1, descending=True, stable=True)
    assert torch.equal(idx.long(), si[:, :k]) and torch.equal(val, sv[:, :k]), (nq, n, k)
print("topk shapes ok")
PY
