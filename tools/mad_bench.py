#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: 64 queries x one MAD-length video (ctx_l ~ 33 k clips, d=512,
window_len=125, top-k 30 => 1 920 windows), stages A->C end to end.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import synth  # noqa: E402
from cone_amd import inference as inf  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx_l", type=int, default=33_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    opt = make_opt("mad", nms_thd=0.5, eval_split_name="test", topk_window=30)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 1).items()})
    ann, vf, qf = synth.make_dataset(opt, args.queries, 1, seed=0, ctx_range=(args.ctx_l, args.ctx_l + 1))
    store = inf.FeatureStore(opt, ann, vf, qf)
    for _ in range(3):
        out, dp = inf.predict_split(model, store, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, dp = inf.predict_split(model, store, opt)       # the call bench.py's config5 times
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": f"MAD-length video: ctx_l={args.ctx_l}, d=512, window_len=125, {args.queries} queries, "
                                  f"top-30 => {dp['n_windows']} windows, stages A-C",
                      "ms_per_step": round(dt * 1e3, 3), "windows_per_s": round(dp["n_windows"] / dt, 1),
                      "queries_per_s": round(args.queries / dt, 1), "n_gpus": 1}))


if __name__ == "__main__":
    main()
