#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: duration of the launches of one kernel and the idle gap in front of them, grouped by the
kernel that ran before.  usage: tools/trace_prev.py <kernel_trace.csv> <substring> [...]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n[:70]


for pat in sys.argv[2:]:
    agg = collections.OrderedDict()
    for prev, cur in zip(rows, rows[1:]):
        if pat not in cur["Kernel_Name"]:
            continue
        blocks = int(cur["Grid_Size_X"]) // int(cur["Workgroup_Size_X"]) * max(1, int(cur["Grid_Size_Y"]) // max(1, int(cur["Workgroup_Size_Y"])))
        key = (blocks, short(prev["Kernel_Name"]), int(prev["Grid_Size_X"]) // int(prev["Workgroup_Size_X"]))
        d = int(cur["End_Timestamp"]) - int(cur["Start_Timestamp"])
        g = int(cur["Start_Timestamp"]) - int(prev["End_Timestamp"])
        pd = int(prev["End_Timestamp"]) - int(prev["Start_Timestamp"])
        agg.setdefault(key, []).append((d, g, pd))
    print(f"== {pat}")
    for (blocks, pname, pblocks), v in sorted(agg.items(), key=lambda kv: -len(kv[1])):
        if len(v) < 20:
            continue
        ds = sorted(x[0] for x in v)
        print(f"  {blocks:5d} workgroups after {pname:70s} ({pblocks:5d} wg, {sum(x[2] for x in v) / len(v) / 1e3:6.1f} us): n={len(v):5d} "
              f"avg {sum(ds) / len(ds) / 1e3:5.2f} us  median {ds[len(ds) // 2] / 1e3:5.2f}  min {ds[0] / 1e3:5.2f}  gap {sum(x[1] for x in v) / len(v) / 1e3:5.2f} us")
