#!/usr/bin/env python3
"""MFMA-busy % and effective clock of the attention kernels from ONE rocprofv3 PMC pass over tools/attn_bench.py:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pm -o m -- python3 tools/attn_bench.py 20000 3
    python3 tools/pmc_quick.py /tmp/pm/.../m_counter_collection.csv"""
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        per[k]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, m in per.items():
    if "attn" not in k: continue
    busy, act, dur = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(m["GRBM_GUI_ACTIVE"]), sum(m["_dur"])
    n = len(m["_dur"])
    print(f"{k[:60]:60s} n={n} avg {dur/n/1e3:8.1f} us  mfma busy {100*(busy/1024)/(act/8):5.1f} %  clock {(act/8)/dur:5.3f} GHz  sq_busy {sum(m['SQ_BUSY_CYCLES'])/act:5.2f}")
