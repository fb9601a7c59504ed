#!/usr/bin/env python3
"""Per-kernel summary of an SQ counter pass (rocprofv3 --pmc SQ_* --kernel-trace --output-format csv).

usage: pmc_sq_summary.py <counter_collection.csv> [clock_MHz=2400] > summary.csv

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration * clock * 1024 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES
counts cycles (MI355X_MICROARCH.md, PMC notes; = 16 x SQ_INSTS_MFMA for V_MFMA_I32_16X16X64_I8, which
the numbers confirm), the chip has 256 CUs x 4 SIMDs.  wait/issue/active are fractions of
SQ_WAVE_CYCLES (disjoint buckets).  Durations are those of the counter run (kernels serialised).
"""
import collections
import csv
import sys

CTRS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]


def main():
    clock = float(sys.argv[2]) if len(sys.argv) > 2 else 2400.0
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        if "mq::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("kernel,workgroups,launches,avg_us,mfma_util,valu_per_mfma,wait_any,wait_inst_any,active_inst_any," + ",".join(CTRS))
    for key, v in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]].values())):
        n = len(v["SQ_WAVE_CYCLES"])
        if n < 10:
            continue
        avg = {c: sum(v[c]) / max(len(v[c]), 1) for c in CTRS}
        us = sum(dur[key].values()) / len(dur[key])
        wave = max(avg["SQ_WAVE_CYCLES"], 1.0)
        mfma = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * clock * 1024.0)
        vpm = avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"] if avg["SQ_INSTS_MFMA"] else 0.0
        print(",".join([f'"{key[0]}"', str(key[1]), str(n), f"{us:.1f}", f"{mfma:.3f}", f"{vpm:.2f}",
                        f"{avg['SQ_WAIT_ANY'] / wave:.3f}", f"{avg['SQ_WAIT_INST_ANY'] / wave:.3f}",
                        f"{avg['SQ_ACTIVE_INST_ANY'] / wave:.3f}"] + [f"{avg[c]:.0f}" for c in CTRS]))


if __name__ == "__main__":
    main()
