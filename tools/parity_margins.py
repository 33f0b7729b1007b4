#!/usr/bin/env python3
"""Print the worst absolute deviation of the HIP path from the reference fixtures (tests/golden/stageB_*.npz):
how much of the 1e-4 tolerance is used -- for the default exact-fp32 path and for the opt-in split_bf16 path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from cone_amd import synth  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

dev = torch.device("cuda", 0)
for name, split in (("stageB_ego4d", 0), ("stageB_ego4d", 1), ("stageB_mad", 0), ("stageB_mad", 1)):
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    opt = make_opt(str(fx["preset"]))
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, int(fx["weight_seed"])).items()})
    model.set_option("split_bf16", split)
    lens_v, lens_q = fx["lens_v"].tolist(), fx["lens_q"].tolist()
    inp = gi.stage_b_inputs(opt, int(fx["input_seed"]), lens_v, lens_q)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = model.forward(t(inp["src_txt"]), t(inp["txt_mask"]), t(inp["src_vid"]), t(inp["vid_mask"]), taps=True)
    Lv = inp["src_vid"].shape[1]
    vm = np.zeros((len(lens_v), Lv + inp["src_txt"].shape[1]), bool)
    for b, (v, q) in enumerate(zip(lens_v, lens_q)):
        vm[b, :v] = True
        vm[b, Lv:Lv + q] = True
    err = {k: float(np.abs(out[k].cpu().numpy() - fx[k]).max()) for k in ("pred_logits", "pred_spans", "hs")}
    err["memory"] = float(np.abs(out["memory"].cpu().numpy() - fx["memory"])[vm].max())
    err["saliency"] = float(np.abs(out["saliency_scores"].cpu().numpy() - fx["saliency_scores"])[vm[:, :Lv]].max())
    print(name, f"split_bf16={split}", {k: f"{v:.2e}" for k, v in err.items()}, "| logits magnitude", f"{np.abs(fx['pred_logits']).max():.1f}")
