#!/usr/bin/env python3
"""Latency of the fused layer tail (out_proj + LN + FFN + LN, ffn.hip) at small row counts: where the 64-row / 4-wave
form (M <= 16 384: at most half of the CUs would hold a 128-row tile) and the 128-row / 8-wave form take over."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ff = 1024
d = lambda t: t.to(dev).contiguous()
W1, b1 = d(torch.randn(ff, 256, generator=g) / 16), d(torch.randn(ff, generator=g) * 0.2)
W2, b2 = d(torch.randn(256, ff, generator=g) / ff ** 0.5), d(torch.randn(256, generator=g) * 0.2)
Wo, bo = d(torch.randn(256, 256, generator=g) / 16), d(torch.randn(256, generator=g) * 0.2)
lg, lb, pg, pb = (d(torch.rand(256, generator=g) + 0.5), d(torch.randn(256, generator=g)),
                  d(torch.rand(256, generator=g) + 0.5), d(torch.randn(256, generator=g) * 0.3))
lib, P = _lib.load(), _lib.ptr
for M in (100, 2000, 9600, 12500, 16384, 16512, 32768, 36000, 40000, 49152, 100000, 265000, 270000):
    X = torch.randn(M, 256, device=dev)
    A = torch.randn(M, 256, device=dev)
    out = torch.empty(M, 256, device=dev)
    call = lambda: _lib.check(lib.cone_test_proj_ffn(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2), P(b2),
                                                     P(lg), P(lb), P(out), M, ff, _lib.stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = M * (4.0 * ff * 256 + 2.0 * 256 * 256)
    print(f"M={M:7d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
