#!/usr/bin/env python3
"""Digests of the pre-filter's outputs (window scores, frame scores, stable top-k) on fixed seeded inputs, for a same-box A/B of
two builds of prefilter.hip (tools/ab_on_box.sh):
    tools/ab_on_box.sh cone_amd/csrc/prefilter.hip tools/probe/_ab/<other>.hip "python3 tools/prefilter_digest.py /tmp/pf_ab.pt"
With a path: the first run saves its outputs there, the next one compares with them case by case (mismatching elements,
largest difference, rank lists)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
h = hashlib.sha1()
path = sys.argv[1] if len(sys.argv) > 1 else None
ref = torch.load(path) if path and os.path.exists(path) else None
out = {}
for ctx_l, dv, W in ((300_017, 512, 125), (44_001, 256, 90), (901, 256, 90), (37, 768, 90), (5_003, 1024, 7)):
    vid = ops.l2_normalize(torch.randn(ctx_l, dv, device=dev, generator=g), 0.0)
    for nq in (1, 2, 3, 4, 5, 7):
        txt = ops.l2_normalize(torch.randn(nq, dv, device=dev, generator=g), 0.0)
        for want_fs in (False, True):
            fs, ws = ops.prefilter_scores(vid, txt, W, frame_scores=want_fs)
            idx, val = ops.topk_windows(ws, min(30, ws.shape[1]))
            key = f"ctx{ctx_l}_dv{dv}_W{W}_q{nq}_fs{int(want_fs)}"
            out[key] = dict(ws=ws.cpu(), idx=idx.cpu(), fs=fs.cpu() if want_fs else None)
            for t in (ws, idx, val) + ((fs,) if want_fs else ()):
                h.update(t.cpu().numpy().tobytes())
            if ref is not None:
                r = ref[key]
                dws = (out[key]["ws"] - r["ws"]).abs()
                msg = f"{key}: window scores {int((dws != 0).sum())}/{dws.numel()} differ (max {float(dws.max()):.2e}), top-k lists " \
                      f"{'equal' if torch.equal(out[key]['idx'], r['idx']) else 'DIFFER'}"
                if want_fs:
                    dfs = (out[key]["fs"] - r["fs"]).abs()
                    msg += f", frame scores {int((dfs != 0).sum())}/{dfs.numel()} differ (max {float(dfs.max()):.2e})"
                if (dws != 0).any() or not torch.equal(out[key]["idx"], r["idx"]):
                    print(msg)
print("prefilter digest", h.hexdigest())
if path and ref is None:
    torch.save(out, path)
