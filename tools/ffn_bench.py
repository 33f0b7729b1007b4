#!/usr/bin/env python3
"""Micro-benchmark of the fused feed-forward kernel (ffn.hip) against linear1 / linear2 as two row-tile GEMMs, on random
operands at the row count of the bench workload (BASELINE configs[1]: ~2.0 M packed token rows).
usage: ffn_bench.py [M] [ff] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2_021_000
ff = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(M, 256, device=dev, generator=g)
W1 = torch.randn(ff, 256, device=dev, generator=g) / 16
b1 = torch.randn(ff, device=dev, generator=g) * 0.1
W2 = torch.randn(256, ff, device=dev, generator=g) / ff ** 0.5
b2 = torch.randn(256, device=dev, generator=g) * 0.1
lg = torch.rand(256, device=dev, generator=g) + 0.5
lb = torch.randn(256, device=dev, generator=g)
out = torch.empty(M, 256, device=dev)
H = torch.empty(M, ff, device=dev)
out2 = torch.empty(M, 256, device=dev)
lib = _lib.load()
P = _lib.ptr
s = _lib.stream()


def fused():
    _lib.check(lib.cone_test_ffn(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out), M, ff, s))


def unfused():
    _lib.check(lib.cone_test_gemm(P(X), None, 0, P(W1), P(b1), None, None, None, P(H), None, None, M, ff, 256, 1, s))
    _lib.check(lib.cone_test_gemm(P(H), None, 0, P(W2), P(b2), P(X), P(lg), P(lb), P(out2), None, None, M, 256, ff,
                                  2 | 4, s))


img = torch.empty(lib.cone_test_ffn_split_image_bytes(ff), dtype=torch.uint8, device=dev)
out3 = torch.empty(M, 256, device=dev)
_packed = [1]


def fused_split():
    _lib.check(lib.cone_test_ffn_split(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out3), M, ff, P(img), _packed[0], s))
    _packed[0] = 0


A = torch.randn(M, 256, device=dev, generator=g)
Wo = torch.randn(256, 256, device=dev, generator=g) / 16
X1 = torch.empty(M, 256, device=dev)


def layer_fused():
    _lib.check(lib.cone_test_proj_ffn(P(A), P(Wo), P(b2), P(X), P(lg), P(lb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb),
                                      P(out), M, ff, s))


wo_img = torch.empty(lib.cone_test_proj_split_image_bytes(), dtype=torch.uint8, device=dev)
_packed2 = [1]


def layer_split():
    _lib.check(lib.cone_test_proj_ffn_split(P(A), P(Wo), P(b2), P(X), P(lg), P(lb), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb),
                                            P(out3), M, ff, P(img), P(wo_img), _packed2[0], s))
    _packed2[0] = 0


def layer_unfused():
    _lib.check(lib.cone_test_gemm(P(A), None, 0, P(Wo), P(b2), P(X), P(lg), P(lb), P(X1), None, None, M, 256, 256, 2 | 4, s))
    _lib.check(lib.cone_test_gemm(P(X1), None, 0, P(W1), P(b1), None, None, None, P(H), None, None, M, ff, 256, 1, s))
    _lib.check(lib.cone_test_gemm(P(H), None, 0, P(W2), P(b2), P(X1), P(lg), P(lb), P(out2), None, None, M, 256, ff,
                                  2 | 4, s))


flops = 4.0 * M * ff * 256
for name, fn in (("fused", fused), ("two GEMMs", unfused), ("fused", fused), ("two GEMMs", unfused),
                 ("fused bf16x3", fused_split), ("fused bf16x3", fused_split),
                 ("proj+ffn", layer_fused), ("3 GEMMs", layer_unfused), ("proj+ffn", layer_fused), ("3 GEMMs", layer_unfused),
                 ("proj+ffn bf16x3", layer_split), ("proj+ffn bf16x3", layer_split)):
    if name.startswith("proj+ffn"):
        flops = 4.0 * M * ff * 256 + 2.0 * M * 256 * 256
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:10s} M={M} ff={ff}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
print("max |fused - two GEMMs| =", float((out - out2).abs().max()))
print("max |proj+ffn bf16x3 - proj+ffn| =", float((out3 - out).abs().max()))
