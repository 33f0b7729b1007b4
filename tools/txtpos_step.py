#!/usr/bin/env python3
"""BASELINE configs[1] without and with --use_txt_pos (cone/config.py:115; text tokens carry a trained position term,
cone/model.py:106): ms per step, one step in flight."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
for txt in (False, True):
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32, use_txt_pos=txt)
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
    print("use_txt_pos" if txt else "default", round((time.perf_counter() - t0) / 6 * 1e3, 2), "ms per step")
