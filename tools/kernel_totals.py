#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 rocpd database (all launches): calls, total ms, average us.  usage: kernel_totals.py results.db [n]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tot, cnt = defaultdict(float), defaultdict(int)
for s, e, n in db.execute("select start, end, name from kernels"):
    k = n.split("(")[0].replace("void ", "").replace("cone::", "")
    tot[k] += (e - s) / 1e6
    cnt[k] += 1
top = sorted(tot.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]
for k, t in top:
    print(f"{t:9.3f} ms  {cnt[k]:5d} calls  {t / cnt[k] * 1e3:9.1f} us  {k[:110]}")
print(f"# total {sum(tot.values()):.3f} ms in {sum(cnt.values())} launches")
