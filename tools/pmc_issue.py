#!/usr/bin/env python3
"""Issue-time model of the step's kernels from two rocprofv3 PMC passes of ONE command (tools/collect_profiles.sh, pass 6):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM \
              --output-format csv -d gpurun_out/pmc_inst -o i -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
              --output-format csv -d gpurun_out/pmc_coexec -o c -- python3 bench.py ...
    python tools/pmc_issue.py gpurun_out profiles/r03_pmc_issue.csv

On gfx950 vector instructions do not hide under the exact-fp32 MFMA (tools/probe/mfma_valu_overlap.hip): a SIMD's time is at
least 32 cycles per v_mfma_f32_16x16x4_f32 + ~4 per other vector instruction + ~8 per transcendental it issues.  Per kernel:
wave-instructions per launch by class, that sum per SIMD (1 024 SIMDs) as `issue_ms` at the launch's effective clock, the
measured duration, and the share of MFMA-busy cycles in which a vector instruction executed too (COEXEC)."""
import csv
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
            per[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return per


def main():
    src, dst = sys.argv[1], sys.argv[2]
    inst = load(f"{src}/pmc_inst/i_counter_collection.csv")
    co = load(f"{src}/pmc_coexec/c_counter_collection.csv")
    rows = []
    for k, m in inst.items():
        if "cone::" not in k:
            continue
        n = len(m["SQ_INSTS_VALU"])
        avg = lambda c: sum(m.get(c, [0.0])) / n
        mfma, valu_all, trans = avg("SQ_INSTS_MFMA"), avg("SQ_INSTS_VALU"), avg("SQ_INSTS_VALU_TRANS_F32")
        lds, salu, vmem = avg("SQ_INSTS_LDS"), avg("SQ_INSTS_SALU"), avg("SQ_INSTS_VMEM")
        valu = max(valu_all - mfma - trans, 0.0)            # SQ_INSTS_VALU counts every vector instruction, MFMAs included
        c = co.get(k, {})
        act, dur = sum(c.get("GRBM_GUI_ACTIVE", [0.0])), sum(c.get("_dur_ns", [0.0]))
        ghz = (act / 8) / dur if dur else 0.0
        if dur and dur / max(1, len(c.get("_dur_ns", [1]))) < 200e3:
            ghz = 0.0       # launches shorter than 200 us: the clock ratio is not a clock (no issue-time figure either)
        busy, coex = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])), sum(c.get("SQ_VALU_MFMA_COEXEC_CYCLES", [0.0]))
        cyc = (32.0 * mfma + 4.0 * valu + 8.0 * trans) / 1024.0        # per SIMD
        issue_ms = cyc / (ghz * 1e9) * 1e3 if ghz else 0.0
        meas_ms = sum(m["_dur_ns"]) / n / 1e6
        rows.append((k, n, mfma, valu, trans, lds, salu, vmem, valu / mfma if mfma else 0.0, issue_ms, meas_ms, ghz,
                     100.0 * coex / busy if busy else 0.0))
    rows.sort(key=lambda r: -r[10] * r[1])
    with open(dst, "w") as f:
        f.write("kernel,launches,mfma_per_launch,other_vector_per_launch,transcendental_per_launch,lds_per_launch,salu_per_launch,"
                "vmem_per_launch,vector_per_mfma,issue_model_ms,measured_ms_profiled,effective_clock_ghz,coexec_pct_of_mfma_busy\n")
        for r in rows:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f},{r[5]:.0f},{r[6]:.0f},{r[7]:.0f},{r[8]:.2f},{r[9]:.3f},"
                    f"{r[10]:.3f},{r[11]:.3f},{r[12]:.1f}\n")
    for r in rows[:8]:
        print(f"{r[0][:52]:52s} x{r[1]:3d}  mfma {r[2]:.3g} vec {r[3]:.3g} ({r[8]:.2f}/mfma) trans {r[4]:.3g}  issue {r[9]:7.3f} ms  "
              f"measured {r[10]:7.3f} ms @ {r[11]:.2f} GHz  coexec {r[12]:.0f} %")


if __name__ == "__main__":
    main()
