#!/usr/bin/env python3
"""Three --use_txt_pos steps of BASELINE configs[1] (for a kernel trace: rocprofv3 --kernel-trace -- python3 tools/txtpos_one.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32, use_txt_pos=True)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=0)
store = inf.FeatureStore(opt, ann, vf, qf)
for _ in range(3):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
