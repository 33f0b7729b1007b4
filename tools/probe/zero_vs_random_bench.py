import os, sys, torch
sys.path.insert(0, "/root/repo")
from cone_amd import _lib
M, ff = 2021000, 1024
dev = torch.device("cuda", 0)
lib = _lib.load(); P = _lib.ptr; s = _lib.stream()
for mode in ("zeros", "random"):
    mk = (lambda *sh: torch.zeros(*sh, device=dev)) if mode == "zeros" else (lambda *sh: torch.randn(*sh, device=dev) * 0.1)
    X, W1, b1, W2, b2 = mk(M, 256), mk(ff, 256), mk(ff), mk(256, ff), mk(256)
    lg, lb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    out = torch.empty(M, 256, device=dev)
    img = torch.empty(lib.cone_test_ffn_split_image_bytes(ff), dtype=torch.uint8, device=dev)
    pk = [1]
    def run_split():
        _lib.check(lib.cone_test_ffn_split(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out), M, ff, P(img), pk[0], s)); pk[0] = 0
    def run_f32():
        _lib.check(lib.cone_test_ffn(P(X), P(W1), P(b1), P(W2), P(b2), P(lg), P(lb), P(out), M, ff, s))
    for name, fn in (("split", run_split), ("fp32", run_f32)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(mode, name, round(e0.elapsed_time(e1) / 10, 3), "ms")
