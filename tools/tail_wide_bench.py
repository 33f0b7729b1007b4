#!/usr/bin/env python3
"""The fused layer tail at FEW rows: time per launch and a digest of the output bits, for both entry points (plain FFN
block, out_proj + LN + FFN block).  Run under tools/ab_variants.sh with -DCONE_FFN_WIDE_GROUPS=0 / =1000000 to compare
the row-owning forms (ffn.hip) with the wide form (ffn_wide.hip): the digests must be equal, the times tell where the wide
form stops paying (the default threshold CONE_FFN_WIDE_GROUPS in ffn.hip)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ff = int(os.environ.get("FF", 1024))
d = lambda t: t.to(dev).contiguous()
W1, b1 = d(torch.randn(ff, 256, generator=g) / 16), d(torch.randn(ff, generator=g) * 0.2)
W2, b2 = d(torch.randn(256, ff, generator=g) / ff ** 0.5), d(torch.randn(256, generator=g) * 0.2)
Wo, bo = d(torch.randn(256, 256, generator=g) / 16), d(torch.randn(256, generator=g) * 0.2)
lg, lb, pg, pb = (d(torch.rand(256, generator=g) + 0.5), d(torch.randn(256, generator=g)),
                  d(torch.rand(256, generator=g) + 0.5), d(torch.randn(256, generator=g) * 0.3))
lib, P = _lib.load(), _lib.ptr
Ms = [int(a) for a in sys.argv[1:]] or [1, 15, 16, 100, 320, 800, 1600, 2048, 3000, 4096, 6000, 8192, 12000, 16384]
Xall = torch.randn(max(Ms), 256, generator=g).to(dev)
Aall = torch.randn(max(Ms), 256, generator=g).to(dev)
scratch = torch.empty(lib.cone_test_proj_ffn_spread_scratch_bytes(ff), dtype=torch.uint8, device=dev)
SPREAD_MAX = int(os.environ.get("SPREAD_MAX", 1024))
for M in Ms:
    X, A = Xall[:M].contiguous(), Aall[:M].contiguous()
    out = torch.empty(M, 256, device=dev)
    calls = {
        "proj": lambda: _lib.check(lib.cone_test_proj_ffn(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2),
                                                          P(b2), P(lg), P(lb), P(out), M, ff, _lib.stream())),
        # the spread form (four launches over single-wave workgroups): only up to its group limit
        "spread": lambda: _lib.check(lib.cone_test_proj_ffn_spread(P(A), P(Wo), P(bo), P(X), P(pg), P(pb), P(W1), P(b1), P(W2),
                                                                   P(b2), P(lg), P(lb), P(out), M, ff, P(scratch), _lib.stream())),
        "ffn": lambda: _lib.check(lib.cone_test_ffn(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out), M, ff,
                                                    _lib.stream())),
    }
    line = f"M={M:6d}:"
    for name, call in calls.items():
        if name == "spread" and M > SPREAD_MAX:
            continue
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        dig = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]
        line += f"  {name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us {dig}"
    print(line)
