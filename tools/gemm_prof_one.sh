#!/bin/bash
# usage: tools/gemm_prof_one.sh <shape> <configs> -> median kernel durations (rocprofv3) for ONE shape filter
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gp; rm -f gpurun_out/gp/t_kernel_trace.csv
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -o t -- python3 tools/gemm_bench.py --only "$1" --configs="$2" ${3:-} > gpurun_out/gp/log 2>&1
python3 - <<'PY'
import collections, csv
rows = list(csv.DictReader(open("gpurun_out/gp/t_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "mq::" not in n or ("gemm" not in n and "splitk" not in n):
        continue
    name = n.split("(")[0].replace("void ", "")
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, blocks), v in agg.items():
    if len(v) < 10:
        continue
    v = sorted(v)
    print(f"   {name:58s} blocks {blocks:5d} n {len(v):3d} median {v[len(v)//2]/1e3:7.1f} us  min {v[0]/1e3:7.1f}")
PY
rm -f gpurun_out/gp/t_kernel_trace.csv
