#!/usr/bin/env python3
"""Micro-benchmark of the fused decoder cross-attention kernels on the bench workload's shape (20 000 windows of
90 clips + 8..20 text tokens, 5 query slots).  usage: dec_cross_bench.py [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
vlen = torch.full((B,), 90, dtype=torch.int32, device=dev)
tlen = torch.randint(8, 21, (B,), device=dev, generator=g, dtype=torch.int32)
L = vlen + tlen
off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
off[1:] = torch.cumsum(L, 0)
M = int(off[-1])
X = torch.randn(M, 256, device=dev, generator=g)
pos = torch.randn(4095, 256, device=dev, generator=g)
DQ = torch.randn(B * 5, 256, device=dev, generator=g)
Wk = torch.randn(256, 256, device=dev, generator=g) / 16
WvT = torch.randn(256, 256, device=dev, generator=g) / 16
bv = torch.randn(256, device=dev, generator=g)
OUT = torch.empty(B * 5, 256, device=dev)
lib = _lib.load()
slabs = torch.empty(lib.cone_test_dec_cross_slab_floats(), device=dev)
P, s = _lib.ptr, _lib.stream()
Lmax = int(L.max())


def run(variant, shared):
    _lib.check(lib.cone_test_dec_cross(P(DQ), P(X), P(pos), P(vlen), P(off), P(Wk), P(WvT), P(bv), P(OUT), B, 5, Lmax,
                                       variant, P(slabs) if shared else None, s))


flops = float((2 * 2 * 40 * L.double() * 256).sum()) + B * 2 * 2 * 5 * 256 * 256      # two contractions + the two folds
for name, v, sh in (("mfma rows-once", 5, False), ("mfma rows-once, shared q", 5, True), ("mfma two-read", 3, False),
                    ("mfma two-read, shared q", 3, True), ("mfma resident", 4, False), ("mfma resident, shared q", 4, True),
                    ("valu", 1, False), ("mfma rows-once", 5, False), ("mfma rows-once, shared q", 5, True)):
    for _ in range(2):
        run(v, sh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(v, sh)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:22s} B={B} M={M}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s algorithmic, {M * 1024 / ms / 1e6:.0f} GB/s of memory rows")
