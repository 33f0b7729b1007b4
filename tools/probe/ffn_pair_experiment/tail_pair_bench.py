#!/usr/bin/env python3
"""The layer tail on ffn_pair.hip's 32x32x2 form against ffn.hip's 16x16x4 form: same bits, time of each.
usage: tail_pair_bench.py [M] [ff] [reps] [proj|ffn|both]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2_064_384      # 63 full rounds of 256 tiles of 128 rows
ff = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
what = sys.argv[4] if len(sys.argv) > 4 else "both"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
X, A = rnd(M, 256), rnd(M, 256)
W1, b1 = rnd(ff, 256) / 16, rnd(ff) * 0.1
W2, b2 = rnd(256, ff) / ff ** 0.5, rnd(256) * 0.1
Wo, bo = rnd(256, 256) / 16, rnd(256) * 0.1
lg, lb, pg, pb = torch.rand(256, device=dev, generator=g) + 0.5, rnd(256), torch.rand(256, device=dev, generator=g) + 0.5, rnd(256)
o_old, o_new = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
lib, P, s = _lib.load(), _lib.ptr, _lib.stream()


def timed(name, fn, flops):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} M={M} ff={ff}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")


if what in ("ffn", "both"):
    img = torch.empty(lib.cone_test_ffn_pair_image_bytes(ff, 0), dtype=torch.uint8, device=dev)
    _lib.check(lib.cone_test_ffn_pair(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_new), M, ff, P(img), 1, s))
    _lib.check(lib.cone_test_ffn(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_old), M, ff, s))
    torch.cuda.synchronize()
    d = (o_old - o_new).abs().max().item()
    print("feed-forward block:", "IDENTICAL BITS" if torch.equal(o_old, o_new) else f"DIFFERENT (max |diff| {d:.3e}, "
          f"{int((o_old != o_new).sum())} of {o_old.numel()} values, finite {bool(torch.isfinite(o_new).all())})")
    fl = 4.0 * M * ff * 256
    for _ in range(2):
        timed("ffn.hip     16x16x4", lambda: _lib.check(lib.cone_test_ffn(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_old), M, ff, s)), fl)
        timed("ffn_pair.hip 32x32x2", lambda: _lib.check(lib.cone_test_ffn_pair(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_new), M, ff, P(img), 0, s)), fl)
if what in ("proj", "both"):
    img = torch.empty(lib.cone_test_ffn_pair_image_bytes(ff, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.cone_test_proj_ffn_pair(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_new), M, ff, P(img), 1, s))
    _lib.check(lib.cone_test_proj_ffn(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_old), M, ff, s))
    torch.cuda.synchronize()
    d = (o_old - o_new).abs().max().item()
    print("layer tail:", "IDENTICAL BITS" if torch.equal(o_old, o_new) else f"DIFFERENT (max |diff| {d:.3e}, "
          f"{int((o_old != o_new).sum())} of {o_old.numel()} values, finite {bool(torch.isfinite(o_new).all())})")
    fl = 4.0 * M * ff * 256 + 2.0 * M * 256 * 256
    for _ in range(2):
        timed("ffn.hip     16x16x4, with out_proj", lambda: _lib.check(lib.cone_test_proj_ffn(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_old), M, ff, s)), fl)
        timed("ffn_pair.hip 32x32x2, with out_proj", lambda: _lib.check(lib.cone_test_proj_ffn_pair(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(o_new), M, ff, P(img), 0, s)), fl)
