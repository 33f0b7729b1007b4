#!/usr/bin/env python3
"""BASELINE configs[1] with a post-norm and a --pre_norm checkpoint (cone/config.py:120): ms per step, one step in flight.
Pre-norm runs the table path with the fused layer tail in its pre-norm form and the folded decoder cross-attention; it has no
first-layer row caches (its in_proj reads norm1(x)) and no first-decoder-layer constants."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
for pre in (False, True):
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32, pre_norm=pre)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
    ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=0)
    store = inf.FeatureStore(opt, ann, vf, qf)
    for _ in range(2): inf.predict_split(model, store, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for _ in range(6):
        h = inf.predict_split_async(model, store, opt)
        if prev: prev.result()
        prev = h
    prev.result(); torch.cuda.synchronize()
    print("pre_norm" if pre else "post_norm", round((time.perf_counter() - t0) / 6 * 1e3, 2), "ms per step")
