#!/bin/bash
# usage: tools/gemm_ab.sh "<variants>" "<shapes>" [configs] [extra gemm_bench flags, e.g. --cold] -> same-box A/B of library variants (mquant_amd/libmquant_hip_<v>.so; "main" = the regular build)
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gp
for s in $2; do
  for v in $1; do
    lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_$v.so; [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip.so
    rm -f gpurun_out/gp/t_kernel_trace.csv
    MQUANT_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -o t -- python3 tools/gemm_bench.py --only "$s" --configs="${3:--1:0}" --tiled ${4:-} > gpurun_out/gp/log 2>&1
    python3 - "$s" "$v" <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open("gpurun_out/gp/t_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "mq::" not in n or ("gemm" not in n and "splitk" not in n):
        continue
    name = n.split("(")[0].replace("void ", "").replace("mq::", "")
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, blocks), v in agg.items():
    if len(v) < 10:
        continue
    v = sorted(v)
    print(f"{sys.argv[1]:12s} {sys.argv[2]:8s} {name:50s} x{blocks:<5d} median {v[len(v)//2]/1e3:7.1f} us  min {v[0]/1e3:7.1f}")
PY
  done
done
rm -f gpurun_out/gp/t_kernel_trace.csv
