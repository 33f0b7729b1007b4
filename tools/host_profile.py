#!/usr/bin/env python3
"""Where the host's share of a bench step goes: time to enqueue the step's launches, wait for the GPU, build the
submission rows (one MI355X, bench workload).  usage: host_profile.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import inference as inf, synth  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(1)
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32, window_batch=32768)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=0)
store = inf.FeatureStore(opt, ann, vf, qf)
for _ in range(2):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
acc = [0.0, 0.0, 0.0, 0.0]
t_all = time.perf_counter()
for _ in range(steps):
    t0 = time.perf_counter()
    dp = inf.device_pipeline(model, store, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows, n = dp["rows"].cpu(), dp["n"].cpu()
    t3 = time.perf_counter()
    inf.format_results(store.ann, opt, rows, n)
    t4 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[i] += d
tot = time.perf_counter() - t_all
print(f"per step: enqueue {acc[0] / steps * 1e3:.2f} ms, GPU wait {acc[1] / steps * 1e3:.2f} ms, D2H {acc[2] / steps * 1e3:.2f} ms, "
      f"format {acc[3] / steps * 1e3:.2f} ms, total {tot / steps * 1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    dp = inf.device_pipeline(model, store, opt)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
