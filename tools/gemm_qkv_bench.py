#!/usr/bin/env python3
"""The two row-GEMM shapes of the step that matter: q|k|v of encoder layer 2 (2.08 M rows x N 768 x K 256, bias) and a
decoder-side N 256 x K 256 GEMM with residual + LayerNorm on 163 840 rows.  Prints ms and TFLOP/s (20 launches each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib, P = _lib.load(), _lib.ptr
for M, N, K, flags in ((2_079_985, 768, 256, 0), (163_840, 256, 256, 2 | 4), (2_079_985, 768, 256, 0), (163_840, 256, 256, 2 | 4)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias, R = torch.randn(N, device=dev), torch.randn(M, N, device=dev) if flags & 2 else None
    lg, lb = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    call = lambda: _lib.check(lib.cone_test_gemm(P(A), None, 0, P(W), P(bias), P(R) if R is not None else None, P(lg), P(lb), P(C),
                                                 None, None, M, N, K, flags, _lib.stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K} flags={flags}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s  checksum {float(C[::4097].double().sum()):.6f}")
    del A, W, R, C
