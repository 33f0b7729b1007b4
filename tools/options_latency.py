#!/usr/bin/env python3
"""Single-query and 8-query step (BASELINE configs[0] shape) for the default model and the reference's other model options:
ms per split, eager.  usage: options_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
for kw in (dict(), dict(pre_norm=True), dict(use_txt_pos=True), dict(num_queries=10), dict(pre_norm=True, use_txt_pos=True)):
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, **kw)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
    line = f"{kw or 'default'}:"
    for nq, nv in ((1, 1), (8, 1)):
        ann, vf, qf = synth.make_dataset(opt, nq, nv, seed=0, ctx_range=(900, 901))
        store = inf.FeatureStore(opt, ann, vf, qf)
        for _ in range(40): out, dp = inf.predict_split(model, store, opt)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): out, dp = inf.predict_split(model, store, opt)
        torch.cuda.synchronize()
        line += f"  {nq} x {nv}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms"
    print(line, flush=True)
