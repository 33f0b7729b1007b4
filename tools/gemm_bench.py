#!/usr/bin/env python3
"""Micro-benchmark of the fp32-MFMA GEMM tiles at the window-model shapes (M = one 4096-window batch).
Interleaved rounds in one process; prints TFLOP/s per shape (median / best)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

SHAPES = [  # name, N, K, flags, a2
    ("qk_proj(A2)", 512, 256, 0, True),
    ("v_proj", 256, 256, 0, False),
    ("out_proj+res+LN", 256, 256, 2 | 4, False),
    ("ffn1+relu", 1024, 256, 1, False),
    ("ffn2+res+LN", 256, 1024, 2 | 4, False),
]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 101
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    bufs = {}
    for name, N, K, flags, a2 in SHAPES:
        A = torch.randn(M, K, device=dev)
        A2 = torch.randn(M, K, device=dev) if a2 else None
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev)
        lg, lb = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        C = torch.empty(M, N, device=dev)
        bufs[name] = (A, A2, W, b, R, lg, lb, C)
    variants = {"tiles-v1": 0x100, "rows-4waves": 0x200, "rows-8waves": 0x300}
    times = {(s[0], v): [] for s in SHAPES for v in variants}
    for r in range(rounds + 1):
        for name, N, K, flags, a2 in SHAPES:
            A, A2, W, b, R, lg, lb, C = bufs[name]
            for vname, vbits in variants.items():
                if vname.startswith("rows") and a2:
                    A2v = None          # the row tile takes a pre-added operand instead of a fused addend
                else:
                    A2v = A2
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.cone_test_gemm(_lib.ptr(A), _lib.ptr(A2v), 0, _lib.ptr(W), _lib.ptr(b),
                                              _lib.ptr(R) if flags & 2 else None, _lib.ptr(lg), _lib.ptr(lb),
                                              _lib.ptr(C), None, None, M, N, K, flags | vbits, _lib.stream()))
                e1.record()
                e1.synchronize()
                if r:
                    times[(name, vname)].append(e0.elapsed_time(e1))
    for vname in variants:
        tot_ms = tot_fl = 0.0
        print(f"--- {vname}")
        for name, N, K, flags, a2 in SHAPES:
            ms = np.array(times[(name, vname)])
            fl = 2.0 * M * N * K
            print(f"{name:18s} N={N:5d} K={K:5d}  median {fl / np.median(ms) / 1e9:7.1f} TF  best {fl / ms.min() / 1e9:7.1f} TF"
                  f"  ({np.median(ms):.3f} ms)")
            tot_ms += np.median(ms)
            tot_fl += fl
        print(f"layer total: {tot_fl / tot_ms / 1e9:.1f} TF  ({tot_ms:.3f} ms per encoder layer at M={M})")


if __name__ == "__main__":
    main()
