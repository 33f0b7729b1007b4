#!/usr/bin/env python3
"""BASELINE.json configs[2]: MAD-scale synthetic long video -- CLIP d=512 features, window_len=125,
~100k windows (ctx_l = 6.2 M clips = 12.7 GB fp32 resident in HBM) -- HBM-bound pre-filter stress.
Reports GB/s of the frame-score stream against the 8 TB/s HBM3E peak (algorithmic bytes
4*ctx_l*dv + Q*4*(dv + num_window): SURVEY.md 8d, the formula bench.py's prefilter_mad object uses).

The product form is timed: window scores with the window max fused into the stream (no (nq, ctx_l) frame-score
matrix), then the stable top-k.  `--queries 1,64` runs several batch sizes in one process (one 12.7 GB allocation) --
the program the rocprofv3 PMC passes of tools/collect_profiles.sh wrap:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o f -- python3 tools/prefilter_bench.py --queries 1,64
"""
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
    ap.add_argument("--queries", type=str, default="1", help="comma-separated query batch sizes")
    ap.add_argument("--topk", type=int, default=30)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--frame_scores", action="store_true", help="also write the (nq, ctx_l) frame-score matrix")
    ap.add_argument("--split_bf16", action="store_true",
                    help="the opt-in three-piece bf16 form (cone_prefilter_scores_split) for >= 8 queries")
    ap.add_argument("--both", action="store_true", help=">= 8 queries: the default fp32 form, then the split form")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    vid = torch.randn(args.ctx_l, args.dv, device=dev, generator=g)
    vid = ops.l2_normalize(vid, 0.0)
    lib = _lib.load()
    nw = ops.num_windows(args.ctx_l, args.W)
    cases = []
    for nq in [int(x) for x in args.queries.split(",")]:
        cases.append((nq, args.split_bf16 and nq >= 8))
        if args.both and nq >= 8:
            cases.append((nq, True))
    txts = {}
    for nq, split in cases:
        if nq not in txts:
            txts[nq] = ops.l2_normalize(torch.randn(nq, args.dv, device=dev, generator=g), 0.0)
        txt = txts[nq]

        def call():
            fs, ws = ops.prefilter_scores(vid, txt, args.W, frame_scores=args.frame_scores, split_bf16=split)
            return fs, ws, ops.topk_windows(ws, args.topk)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        lib.cone_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fs, ws, (idx, val) = call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        buf = np.zeros((4096, 5))
        n = lib.cone_prof_collect(buf.ctypes.data, 4096)
        lib.cone_prof_enable(0)
        rec = buf[:n]
        per_step_ms = rec[np.isin(rec[:, 0], (0, 1, 2, 4, 5))][:, 4].sum() / args.steps
        alg = 4.0 * args.ctx_l * args.dv + nq * 4.0 * (args.dv + nw)
        # reference check: window scores of the first windows from a torch product on the rows they cover
        S = args.W // 2
        nchk = 64
        ref_fs = (vid[:(nchk - 1) * S + args.W] @ txt.t()).t()
        ref_ws = ref_fs[:, :(nchk - 2) * S + args.W].unfold(1, args.W, S).max(dim=2).values
        err = float((ws[:, 1:1 + ref_ws.shape[1]] - ref_ws).abs().max())
        print(json.dumps({
            "workload": f"MAD-scale pre-filter: ctx_l={args.ctx_l}, d={args.dv}, window_len={args.W}, {nw} windows, "
                        f"{nq} query(ies), top-{args.topk}, frame-score matrix {'written' if args.frame_scores else 'not written'}"
                        + (", split_bf16 (three-piece bf16 operands)" if split else ""),
            "ms_per_query_batch": round(dt * 1e3, 3), "windows_per_s": round(nw * nq / dt, 1),
            "frame_score_ms": round(per_step_ms, 3),
            "roofline": {"bound": "hbm", "achieved": round(alg / (per_step_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(alg / (per_step_ms * 1e-3) / 8e12, 4)},
            "path_frac": round(alg / dt / 8e12, 4), "max_abs_err_window_scores_vs_torch": err}))
        del fs, ws


if __name__ == "__main__":
    main()
