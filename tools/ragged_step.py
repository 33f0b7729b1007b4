#!/usr/bin/env python3
"""K steps of the headline workload on the RAGGED split of bench.py's config2_ragged (1 000 queries x 50 videos, ctx_l ~
U[200, 1500)) -- the program tools/collect_profiles.sh wraps in rocprofv3 to list the GPU-idle gaps of a ragged step."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=11, ctx_range=(200, 1500))
store = inf.FeatureStore(opt, ann, vf, qf)
for _ in range(2):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    _, dp = inf.predict_split(model, store, opt)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "config2_ragged", "n_windows": dp["n_windows"], "ms_per_step": round(dt * 1e3, 3), "steps": steps}))
