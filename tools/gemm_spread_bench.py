#!/usr/bin/env python3
"""Row GEMM at a handful of row groups: the spread form (one wave per 16 x 16 output tile) against the workgroup-per-16-rows
form, by row count and output width -- where the spread form stops paying (its single-wave workgroups re-read the A and W
slabs once per output tile).  usage: gemm_spread_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib, P = _lib.load(), _lib.ptr
NO_SPREAD = 8
for N, K in ((256, 256), (768, 256), (256, 768), (256, 512)):
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    for M in (16, 100, 160, 320, 512, 640, 900, 1024):
        A = torch.randn(M, K, device=dev)
        C = torch.empty(M, N, device=dev)
        res = []
        for fl in (0, NO_SPREAD):
            call = lambda: _lib.check(lib.cone_test_gemm(P(A), None, 0, P(W), P(bias), None, None, None, P(C), None, None, M, N, K,
                                                         fl, _lib.stream()))
            for _ in range(5):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"N={N} K={K} M={M:5d} ({(M + 15) // 16:3d} groups, {(M + 15) // 16 * N // 16:5d} tiles): spread {res[0]:6.1f} us   16-row form {res[1]:6.1f} us")
