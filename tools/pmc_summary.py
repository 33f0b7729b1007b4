#!/usr/bin/env python3
"""Summarise three rocprofv3 PMC passes of ONE command (FETCH_SIZE / WRITE_SIZE / MFMA busy, each collected in its own
run as /opt/skills/guides/MI355X_MICROARCH.md prescribes) into a CSV + the JSON bench.py reads for `roofline.traffic`.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv \
              -d gpurun_out/pmc_mfma -o m -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out profiles/r03_pmc                     # -> r03_pmc_counters.csv, r03_pmc_traffic.json
    python tools/pmc_summary.py gpurun_out profiles/r03_pmc_prefilter _pf       # passes in pmc_fetch_pf/ ... -> r03_pmc_prefilter.json

Corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the 128-B requests of
a wide coalesced stream at 64 B, so reads are doubled -- for EVERY kernel (round 2 exempted the fused layer tail on a
mis-calibration: its raw 2.24 GB per encoder launch is below the 3.96 GB its two (M, 256) inputs weigh, which a read
count cannot be; doubled it is 1.13 x algorithmic); WRITE_SIZE is exact for 16-B-per-lane stores.
MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

READ_FACTOR = 2.0
MIN_CLOCK_US = 200.0        # the effective clock of a launch is GRBM_GUI_ACTIVE / its wall time: meaningless for short kernels
#                             (the counter runs on while the launch drains; 3.5 - 12 "GHz" came out for anything < 50 us)


def csrc_hashes():
    """sha1 of every kernel source (cone_amd/csrc): the stamp bench.py compares before it quotes a table as `traffic`."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cone_amd", "csrc")
    return {f: hashlib.sha1(open(os.path.join(d, f), "rb").read()).hexdigest()[:12]
            for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h", ".c"))}


def load(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE"):          # one row per dispatch: its wall time rides along
            per[k]["_dur_ns_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return per


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sfx = sys.argv[3] if len(sys.argv) > 3 else ""
    f = load(f"{src}/pmc_fetch{sfx}/f_counter_collection.csv")
    w = load(f"{src}/pmc_write{sfx}/w_counter_collection.csv")
    m = load(f"{src}/pmc_mfma{sfx}/m_counter_collection.csv")
    rows, traffic = [], {}
    for k in sorted(f, key=lambda k: -sum(f[k]["FETCH_SIZE"])):
        if "cone::" not in k:
            continue
        fv, wv = f[k]["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", [0.0])
        mm = m.get(k, {})
        busy = sum(mm.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
        act = sum(mm.get("GRBM_GUI_ACTIVE", [0.0]))
        rd = READ_FACTOR * 1024 * sum(fv) / len(fv)
        wr = 1024 * sum(wv) / len(wv)
        util = 100.0 * (busy / 1024) / (act / 8) if act else 0.0
        durs = mm.get("_dur_ns_GRBM_GUI_ACTIVE", [])
        dur = sum(durs)
        avg_us = dur / max(1, len(durs)) / 1e3
        # effective shader clock under the profiler (guide: DVFS) -- only where a launch is long enough for the ratio to mean it
        ghz = (act / 8) / dur if dur and avg_us >= MIN_CLOCK_US else None
        rows.append((k, len(fv), rd, wr, rd + wr, util, avg_us, ghz))
        traffic[k] = {"launches": len(fv), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                      "hbm_bytes_per_launch": rd + wr, "fetch_size_raw_bytes_per_launch": rd / READ_FACTOR,
                      "mfma_busy_pct": round(util, 1), "avg_us_profiled": round(avg_us, 1),
                      "effective_clock_ghz": None if ghz is None else round(ghz, 3),
                      "hbm_gbs_profiled": round((rd + wr) / (avg_us * 1e-6) / 1e9, 1) if avg_us else None}
    if dst.endswith("_prefilter"):
        out_csv, out_json = dst + "_counters.csv", dst + ".json"
    else:
        out_csv, out_json = dst + "_counters.csv", dst + "_traffic.json"
    with open(out_csv, "w") as o:
        o.write("kernel,launches,hbm_read_bytes_per_launch(2 x FETCH_SIZE),hbm_write_bytes_per_launch,hbm_bytes_per_launch,"
                "mfma_busy_pct,avg_us_profiled,effective_clock_ghz\n")
        for r in rows:
            clk = "" if r[7] is None else f"{r[7]:.3f}"
            o.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f},{r[5]:.1f},{r[6]:.1f},{clk}\n")
        o.write("# effective_clock_ghz only for launches of >= %.0f us; collected on cone_amd/csrc %s\n"
                % (MIN_CLOCK_US, " ".join(f"{k}@{v}" for k, v in csrc_hashes().items())))
    traffic["_csrc"] = csrc_hashes()        # what the counters were collected on (bench.py: another revision => traffic null)
    with open(out_json, "w") as o:
        json.dump(traffic, o, indent=1, sort_keys=True)
    print(open(out_csv).read())


if __name__ == "__main__":
    main()
