#!/usr/bin/env python3
"""Randomised soak of the pre-filter entry alone (cone_prefilter_scores + cone_topk_windows): random video lengths (1 clip up),
feature widths, window lengths (even and odd), 1 .. 70 queries -- every kernel form (streaming 1 / 2 / 4 queries, 16 / 32 / 64-query
matrix-core tiles, with and without the frame-score matrix) against float64 on the host: frame scores, window scores = max over
the window's frames (cone/inference.py:284-295), rank list = the stable descending order of the returned scores.
usage: fuzz_prefilter.py [iterations] [seed0]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
worst_fs = worst_ws = 0.0
t0 = time.time()
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    dv = int(rng.choice([256, 512, 768, 1024]))
    W = int(rng.choice([2, 3, 7, 8, 64, 90, 125, 126, 255]))
    ctx_l = int(rng.choice([1, 2, W // 2 + 1, W, W + 1, 3 * W + 5, 1000, 4097, 20011]))
    nq = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 70]))
    g = torch.Generator().manual_seed(seed0 + it)
    vid = torch.randn(ctx_l, dv, generator=g)
    txt = torch.randn(nq, dv, generator=g)
    vid, txt = vid / vid.norm(dim=1, keepdim=True), txt / txt.norm(dim=1, keepdim=True)
    ref_fs = (txt.double() @ vid.double().T)                              # (nq, ctx_l)
    S = W // 2
    nw = -(-ctx_l // S) + 1
    ref_ws = torch.stack([ref_fs[:, max((i - 1) * S, 0):min((i - 1) * S + W, ctx_l)].max(dim=1).values for i in range(nw)], 1)
    dvid, dtxt = vid.to(dev).contiguous(), txt.to(dev).contiguous()
    for want_fs in (False, True):
        fs, ws = ops.prefilter_scores(dvid, dtxt, W, frame_scores=want_fs)
        tag = f"iter {it} (seed {seed0 + it}): ctx_l {ctx_l} dv {dv} W {W} nq {nq} fs {want_fs}"
        assert tuple(ws.shape) == (nq, nw), tag
        e_ws = float((ws.double().cpu() - ref_ws).abs().max())
        assert e_ws < 2e-6, (tag, e_ws)
        worst_ws = max(worst_ws, e_ws)
        if want_fs:
            e_fs = float((fs.double().cpu() - ref_fs).abs().max())
            assert e_fs < 2e-6, (tag, e_fs)
            worst_fs = max(worst_fs, e_fs)
            # the fused window max is the max over the STORED frame scores, bit for bit
            mine = torch.stack([fs[:, max((i - 1) * S, 0):min((i - 1) * S + W, ctx_l)].max(dim=1).values for i in range(nw)], 1)
            assert torch.equal(mine, ws), tag
        k = min(int(rng.choice([1, 5, 30, nw])), nw)           # (cone_topk_windows takes k <= num_window; the batched entry pads)
        idx, val = ops.topk_windows(ws, k)
        order = torch.sort(ws.cpu(), dim=1, descending=True, stable=True)[1][:, :k]
        assert torch.equal(idx.cpu().long(), order), tag
print(f"prefilter fuzz ok: {iters} random shapes in {time.time() - t0:.0f} s; worst |frame score - fp64| {worst_fs:.2e}, "
      f"worst |window score - fp64| {worst_ws:.2e}")
