#!/usr/bin/env python3
"""Rank r's share of the 8-rank window-sharded split (BASELINE configs[3]) replayed on one GPU, no collective
(cone_amd.parallel virtual ranks): ms per step; wrap in rocprofv3 --kernel-trace for the per-kernel picture.
usage: proxy_bench.py [world] [rank] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, parallel as par, synth  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.set_num_threads(1)
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=0)
store = inf.FeatureStore(opt, ann, vf, qf)
start = lambda: par.predict_split_distributed_async(model, store, opt, mode="window", format_shard=True, virtual=(rank, world))


def run(n):     # one step in flight, like bench.py's N > 1 headline
    prev = None
    for _ in range(n):
        h = start()
        if prev is not None:
            prev.result()
        prev = h
    prev.result()


run(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"world": world, "rank": rank, "ms_per_step": round(dt * 1e3, 3)}))
