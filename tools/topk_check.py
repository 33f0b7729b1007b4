#!/usr/bin/env python3
"""Two-level stable top-k (cone_topk_windows_ws) against torch.sort(stable=True) on adversarial rows: ties everywhere,
sorted rows (survivors of the threshold selection cluster in few lanes -> its fallback), constant rows, -inf, k = 64 / 65
(fast path / pass-based path), rows barely over the two-level threshold.  Test infrastructure; run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cone_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
n_ok = 0
for nq, n, k in ((3, 100001, 30), (2, 9000, 256), (2, 9000, 300), (5, 50000, 1), (1, 8193, 64), (4, 70000, 200), (2, 300000, 30),
                 (2, 20000, 64), (2, 20000, 65), (3, 12345, 20), (1, 8193, 2)):
    rows = {
        "randint": torch.randint(0, 50, (nq, n), generator=g).float(),
        "randn": torch.randn(nq, n, generator=g),
        "ascending": torch.arange(n, dtype=torch.float32).repeat(nq, 1),
        "descending": torch.arange(n, 0, -1, dtype=torch.float32).repeat(nq, 1),
        "constant": torch.full((nq, n), 3.0),
        "blocks": (torch.arange(n) // 4096).float().repeat(nq, 1),
        "one_hot_chunk": torch.cat([torch.zeros(nq, n - 4096), torch.randn(nq, 4096, generator=g) + 10], 1) if n > 4096 else None,
        "neg_inf": torch.where(torch.rand(nq, n, generator=g) < 0.999, torch.tensor(float("-inf")), torch.randn(nq, n, generator=g)),
    }
    for name, x in rows.items():
        if x is None:
            continue
        x = x.to(dev).contiguous()
        idx, val = ops.topk_windows(x, k)
        sv, si = torch.sort(x, dim=1, descending=True, stable=True)
        assert torch.equal(idx.long(), si[:, :k]) and torch.equal(val, sv[:, :k]), (name, nq, n, k)
        n_ok += 1
print(f"topk ok: {n_ok} cases")
