"""Per-tile fixed cost vs per-chunk cost of ffn_split_kernel<false>: time at several dim_feedforward."""
import sys, torch
sys.path.insert(0, "/root/repo")
from cone_amd import _lib
M = 2021000
dev = torch.device("cuda", 0)
lib = _lib.load(); P = _lib.ptr; s = _lib.stream()
X = torch.randn(M, 256, device=dev) * 0.5
lg, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
out = torch.empty(M, 256, device=dev)
res = []
for ff in (64, 256, 512, 1024, 2048):
    W1, b1, W2, b2 = torch.randn(ff, 256, device=dev) / 16, torch.zeros(ff, device=dev), torch.randn(256, ff, device=dev) / ff ** 0.5, torch.zeros(256, device=dev)
    img = torch.empty(lib.cone_test_ffn_split_image_bytes(ff), dtype=torch.uint8, device=dev)
    pk = [1]
    def run():
        _lib.check(lib.cone_test_ffn_split(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out), M, ff, P(img), pk[0], s)); pk[0] = 0
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res.append((ff, ms))
    print(f"ff={ff}: {ms:.3f} ms  ({ms * 1e3 / 61.7:.1f} us per tile, {ms * 1e3 / 61.7 / (ff / 32):.2f} us per 32-unit chunk)")
(f0, t0), (f1, t1) = res[0], res[-1]
per_chunk = (t1 - t0) / ((f1 - f0) / 32)
print(f"per chunk {per_chunk * 1e3 / 61.7:.2f} us, fixed per tile {(t0 - per_chunk * f0 / 32) * 1e3 / 61.7:.1f} us")
