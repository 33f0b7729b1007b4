#!/usr/bin/env python3
"""Micro-benchmark of the encoder self-attention kernel on the bench workload's shape (20 000 windows of 90 clips +
8..20 text tokens, 8 heads x 32).  usage: attn_bench.py [B] [reps]"""
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
QKV = torch.randn(M, 768, device=dev, generator=g)
pos = torch.randn(4096, 512, device=dev, generator=g)
pos[4095] = 0          # the table's zero row (a text token's position term)
OUT = torch.empty(M, 256, device=dev)
# gather mode: 50 videos of 900 clips, windows at stride 45; 1000 queries' text rows
qkv_vid = torch.randn(50 * 900, 768, device=dev, generator=g)
vrow0 = (torch.randint(0, 50, (B,), device=dev, generator=g) * 900 + torch.randint(0, 18, (B,), device=dev, generator=g) * 45).to(torch.int32)
qkv_txt = torch.randn(1000 * 20, 768, device=dev, generator=g)
trow0 = (torch.randint(0, 1000, (B,), device=dev, generator=g) * 20).to(torch.int32)
lib = _lib.load()
P, s = _lib.ptr, _lib.stream()
Lmax = int(L.max())


def run(mode):
    _lib.check(lib.cone_test_enc_attn(mode, P(QKV), P(qkv_vid), P(qkv_txt), P(pos), P(vrow0), P(vlen), P(trow0), P(off),
                                      P(OUT), B, Lmax, 4095, s))


flops = float((4.0 * L.double() ** 2 * 256).sum())
ref = {}
CASES = (("packed", 0), ("gather", 1), ("pos-add", 2), ("packed/wave", 0x200), ("gather/wave", 0x201), ("pos-add/wave", 0x202),
         ("packed", 0), ("gather", 1), ("pos-add", 2))
if os.environ.get("ATTN_CASES"):                  # e.g. ATTN_CASES=0,2,256
    CASES = tuple((f"mode {int(c):#x}", int(c)) for c in os.environ["ATTN_CASES"].split(","))
for name, mode in CASES:
    for _ in range(2):
        run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    same = ""
    if mode & 0xff in ref:
        same = "  same bits as the other form" if torch.equal(ref[mode & 0xff], OUT) else "  DIFFERENT from the other form"
    ref.setdefault(mode & 0xff, OUT.clone())
    print(f"{name:10s} B={B} M={M}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s algorithmic{same}")
