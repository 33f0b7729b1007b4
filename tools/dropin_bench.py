#!/usr/bin/env python3
"""The drop-in entry on its own (bench.py's `dropin_forward` and `localizer` extras), for profiling:
    rocprofv3 --kernel-trace --stats -d out -o t -- python3 tools/dropin_bench.py [dropin|localizer] [steps]
    python3 tools/dropin_trace.py out/t_results.db      # launch sequence of the last batch"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cone_amd import inference as inf, synth  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "dropin"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if what == "localizer":
    print(json.dumps(bench.bench_localizer(steps=steps)))
else:
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
    for kv in os.environ.get("CONE_SET_OPTION", "").split():
        name, _, val = kv.partition("=")
        model.set_option(name, int(val))
    ann, vf, qf = synth.make_dataset(opt, 320, 16, seed=0)
    store = inf.FeatureStore(opt, ann, vf, qf)
    print(json.dumps(bench.bench_dropin_forward(model, store, opt, None, steps=steps)))
