#!/usr/bin/env python3
"""Row GEMM, small-M form (16-row tiles) against the 128-row tile by row count: where the crossover is
(build with CONE_HIPCC_FLAGS=-DCONE_RS_MAX_WGS=1048576 to force the small form at every size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib, P = _lib.load(), _lib.ptr
for N, K in ((256, 256), (768, 256), (256, 512)):
    W = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    for M in (100, 1024, 2048, 4096, 8192, 12500, 16384, 32768, 65536):
        A = torch.randn(M, K, device=dev)
        C = torch.empty(M, N, device=dev)
        res = []
        for fam in (0, 0x300):
            call = lambda: _lib.check(lib.cone_test_gemm(P(A), None, 0, P(W), P(bias), None, None, None, P(C), None, None, M, N, K,
                                                         1 | fam, _lib.stream()))
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"N={N} K={K} M={M:6d}: auto {res[0]:7.1f} us   128-row tile {res[1]:7.1f} us")
