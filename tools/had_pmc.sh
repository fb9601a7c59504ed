#!/bin/bash
# usage: tools/had_pmc.sh [HAD_FAST=0|1] -> SQ instruction-mix counters of the Hadamard kernels (two rocprofv3 --pmc passes, kernels serialised)
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export HAD_FAST=${1:-0}
export HAD_SHAPES=${2:-llm.down}
rm -rf gpurun_out/hpmc; mkdir -p gpurun_out/hpmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"
P2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d gpurun_out/hpmc -o a -- python3 tools/had_bench.py > gpurun_out/hpmc/log_a 2>&1 || tail -5 gpurun_out/hpmc/log_a
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d gpurun_out/hpmc -o b -- python3 tools/had_bench.py > gpurun_out/hpmc/log_b 2>&1 || tail -5 gpurun_out/hpmc/log_b
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/hpmc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "hadamard" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["Workgroup_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, v in agg.items():
    us = sum(dur[key]) / len(dur[key])
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    wave = avg.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{key[0]} wg={key[1]} threads={key[2]} avg {us:.1f} us (serialised, profiled)")
    for c in sorted(avg):
        extra = f"  ({avg[c] / wave:.3f} of WAVE_CYCLES)" if c.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        print(f"    {c:28s} {avg[c]:14.0f}{extra}")
PY
rm -f gpurun_out/hpmc/*_counter_collection.csv gpurun_out/hpmc/*_kernel_trace.csv
