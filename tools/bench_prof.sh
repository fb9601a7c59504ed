#!/bin/bash
# usage: tools/bench_prof.sh <out-prefix> [bench flags] -> rocprofv3 --kernel-trace --stats of bench.py, our kernels per (name, grid)
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=$1; shift
rm -rf gpurun_out/bp; mkdir -p gpurun_out/bp $(dirname $OUT)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bp -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-prefill --no-secondary --no-floor-model "$@" > ${OUT}_bench.json 2> gpurun_out/bp/err
python3 - "$OUT" <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open("gpurun_out/bp/b_kernel_trace.csv")))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "mq::" not in n or "mfma_burn" in n:       # (the bench's own sustained-rate probe is not part of the step)
        continue
    name = n.split("(")[0].replace("void ", "")
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# one tile shape serves two layer shapes of very different depth (down_proj / o_proj, ViT fc2 / proj): split a key whose
# durations fall into two clusters (largest relative gap > 1.5x with >= 15 % of the launches on either side)
split = collections.OrderedDict()
for (name, blocks), v in agg.items():
    sv = sorted(v)
    best, cut = 1.0, None
    for i in range(max(1, len(sv) * 15 // 100), len(sv) - max(1, len(sv) * 15 // 100)):
        if sv[i] / sv[i - 1] > best:
            best, cut = sv[i] / sv[i - 1], i
    if cut is not None and best > 1.5:
        split[(name + " [shorter launches]", blocks)] = sv[:cut]
        split[(name + " [longer launches]", blocks)] = sv[cut:]
    else:
        split[(name, blocks)] = v
agg = split
tot = sum(sum(v) for v in agg.values())
with open(sys.argv[1] + "_kernel_stats.csv", "w") as fh:
    fh.write("kernel,workgroups,calls,avg_us,min_us,total_ms,share\n")
    for (name, blocks), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fh.write(f'"{name}",{blocks},{len(v)},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{sum(v)/1e6:.3f},{sum(v)/tot:.4f}\n')
print(open(sys.argv[1] + "_kernel_stats.csv").read())
PY
[ -n "${KEEP_TRACE:-}" ] || rm -f gpurun_out/bp/b_kernel_trace.csv
