#!/usr/bin/env python3
"""Summarise the three rocprofv3 PMC passes of bench.py (FETCH_SIZE / WRITE_SIZE / MFMA busy, each collected
in its own run as /opt/skills/guides/MI355X_MICROARCH.md prescribes) into a CSV + the JSON bench.py reads for
`roofline.traffic`.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv \
              -d gpurun_out/pmc_mfma -o m -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out profiles/r01_pmc

Corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B
requests at 64 B, so reads are doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":          # one row per dispatch: its wall time rides along
            per[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return per


# Read-side correction per kernel (guide, section HBM: the x2 holds for wide coalesced 16-B-per-lane streams that the L2
# requests as 128-B lines; "other access widths are uncalibrated: calibrate on a known byte count in your own access
# pattern").  ffn_fused_kernel reads its row tiles as 64-B row segments (4 lanes x 16 B per row and instruction): its
# fabric requests are 64 B and FETCH_SIZE is exact -- calibrated on the encoder launches, whose algorithmic read is the
# two (M, 256) fp32 inputs = 4.14 GB at M = 2.02 M rows against 2 x 1.8 GB counted (weights are served by the L2).
READ_FACTOR = {"cone::ffn_fused_kernel<true, false>": 1.0, "cone::ffn_fused_kernel<false, false>": 1.0,
               "cone::ffn_fused_kernel<true, true>": 1.0}

src, dst = sys.argv[1], sys.argv[2]
f = load(f"{src}/pmc_fetch/f_counter_collection.csv")
w = load(f"{src}/pmc_write/w_counter_collection.csv")
m = load(f"{src}/pmc_mfma/m_counter_collection.csv")
rows, traffic = [], {}
for k in sorted(f, key=lambda k: -sum(f[k]["FETCH_SIZE"])):
    if "cone::" not in k:
        continue
    fv, wv = f[k]["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", [0.0])
    mm = m.get(k, {})
    busy = sum(mm.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
    act = sum(mm.get("GRBM_GUI_ACTIVE", [0.0]))
    rd = READ_FACTOR.get(k, 2.0) * 1024 * sum(fv) / len(fv)
    wr = 1024 * sum(wv) / len(wv)
    util = 100.0 * (busy / 1024) / (act / 8) if act else 0.0
    dur = sum(mm.get("_dur_ns", [0.0]))
    ghz = (act / 8) / dur if dur else 0.0                   # effective shader clock under the profiler (guide: DVFS)
    avg_us = dur / max(1, len(mm.get("_dur_ns", []))) / 1e3
    rows.append((k, len(fv), rd, wr, rd + wr, util, avg_us, ghz))
    traffic[k] = {"launches": len(fv), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr, "mfma_busy_pct": round(util, 1),
                  "avg_us_profiled": round(avg_us, 1), "effective_clock_ghz": round(ghz, 3)}
with open(dst + "_counters.csv", "w") as o:
    o.write("kernel,launches,hbm_read_bytes_per_launch(FETCH_SIZE x READ_FACTOR),hbm_write_bytes_per_launch,hbm_bytes_per_launch,"
            "mfma_busy_pct,avg_us_profiled,effective_clock_ghz\n")
    for r in rows:
        o.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f},{r[5]:.1f},{r[6]:.1f},{r[7]:.3f}\n")
with open(dst + "_traffic.json", "w") as o:
    json.dump(traffic, o, indent=1, sort_keys=True)
print(open(dst + "_counters.csv").read())
