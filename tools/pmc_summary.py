#!/usr/bin/env python3
"""Average rocprofv3 counter_collection.csv values per counter for kernels matching argv[2]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "mq::"
agg = {}
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
