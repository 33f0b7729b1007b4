#!/usr/bin/env python3
"""BASELINE.json configs[2]: MAD-scale synthetic long video -- CLIP d=512 features, window_len=125,
~100k windows (ctx_l = 6.2 M clips = 12.7 GB fp32 resident in HBM) -- HBM-bound pre-filter stress.
Reports GB/s of the frame-score stream against the 8 TB/s HBM3E peak (algorithmic bytes
4*ctx_l*dv + Q*4*(dv + num_window): SURVEY.md 8d, the formula bench.py's prefilter_mad object uses)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx_l", type=int, default=6_200_000)
    ap.add_argument("--dv", type=int, default=512)
    ap.add_argument("--W", type=int, default=125)
    ap.add_argument("--queries", type=int, default=1)
    ap.add_argument("--topk", type=int, default=30)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    vid = torch.randn(args.ctx_l, args.dv, device=dev, generator=g)
    vid = ops.l2_normalize(vid, 0.0)
    txt = ops.l2_normalize(torch.randn(args.queries, args.dv, device=dev, generator=g), 0.0)
    lib = _lib.load()
    nw = ops.num_windows(args.ctx_l, args.W)
    for _ in range(2):
        fs, ws = ops.prefilter_scores(vid, txt, args.W)
        idx, val = ops.topk_windows(ws, args.topk)
    torch.cuda.synchronize()
    lib.cone_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fs, ws = ops.prefilter_scores(vid, txt, args.W)
        idx, val = ops.topk_windows(ws, args.topk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    buf = np.zeros((4096, 5))
    n = lib.cone_prof_collect(buf.ctypes.data, 4096)
    lib.cone_prof_enable(0)
    rec = buf[:n]
    fs_ms = rec[np.isin(rec[:, 0], (0, 1, 2, 4, 5))][:, 4]
    per_step_ms = fs_ms.sum() / args.steps
    alg = 4.0 * args.ctx_l * args.dv + args.queries * 4.0 * (args.dv + nw)
    # reference check on a slice
    ref = (vid[:4096] @ txt.t()).t()
    err = float((fs[:, :4096] - ref).abs().max())
    out = {"workload": f"MAD-scale pre-filter: ctx_l={args.ctx_l}, d={args.dv}, window_len={args.W}, {nw} windows, "
                       f"{args.queries} query(ies), top-{args.topk}",
           "ms_per_query_batch": round(dt * 1e3, 3), "windows_per_s": round(nw * args.queries / dt, 1),
           "frame_score_ms": round(per_step_ms, 3),
           "roofline": {"bound": "hbm", "achieved": round(alg / (per_step_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(alg / (per_step_ms * 1e-3) / 8e12, 4)},
           "max_abs_err_vs_torch": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
