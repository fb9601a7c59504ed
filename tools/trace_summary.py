#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) count / avg / total."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
only = sys.argv[2] if len(sys.argv) > 2 else "mq::"
agg = collections.OrderedDict()
for r in rows:
    if only not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':62s} {'blocks':>6s} {'vgpr':>4s} {'agpr':>4s} {'lds':>6s} {'scr':>4s} {'n':>5s} {'avg_us':>8s} {'min_us':>8s} {'total_ms':>9s}")
for k, v in agg.items():
    print(f"{k[0]:62s} {k[1]:6d} {k[2]:>4s} {k[3]:>4s} {k[4]:>6s} {k[5]:>4s} {len(v):5d} {sum(v)/len(v)/1e3:8.1f} {min(v)/1e3:8.1f} {sum(v)/1e6:9.2f}")
