#!/bin/bash
# Round 5, VERDICT r4 item 1a: every clock reading of the GEMM launches in one run, one box.  Output: gpurun_out/r5clock/*.txt
#   1. tools/probes/clock_recon.bin: register-only int8 MFMA burns (zeros / constant / random operand data) with in-kernel
#      s_memtime vs s_memrealtime, rocm-smi beside them, and the single-wave meter alone and concurrent;
#   2. the same binary under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES (separate pass);
#   3. tools/clock_recon.py: the product's gate|up and down_proj launches (stamp builds), four kinds of operand data;
#   4. gate|up, down_proj and the exact Hadamard under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES;
#   (capping the shader clock as a control experiment is not possible: the pool refuses every job that changes a device setting.)
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5clock; rm -rf $O; mkdir -p $O
echo "--- 1. probe"; timeout 300 tools/probes/clock_recon.bin 2.5 > $O/probe.txt 2>&1; echo "rc=$?"; tail -3 $O/probe.txt
echo "--- 3. product GEMMs"; timeout 600 python3 tools/clock_recon.py > $O/gemm.txt 2>&1; echo "rc=$?"; tail -12 $O/gemm.txt
pmc_pass() {   # name, command...
  local name=$1; shift
  rm -rf gpurun_out/pmcc; mkdir -p gpurun_out/pmcc
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES GRBM_COUNT --kernel-trace --output-format csv -d gpurun_out/pmcc -o t -- "$@" > gpurun_out/pmcc/log 2>&1 || { echo "pmc pass $name failed"; tail -5 gpurun_out/pmcc/log; }
  python3 - "$name" <<'PY' >> gpurun_out/r5clock/pmc.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmcc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"== pass {sys.argv[1]}")
for k, v in agg.items():
    us = sum(dur[k].values()) / len(dur[k])
    if us < 8: continue
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    g, s, n = avg.get("GRBM_GUI_ACTIVE", 0), avg.get("SQ_BUSY_CYCLES", 0), avg.get("GRBM_COUNT", 0)
    print(f"  {k[0]:60s} wg {k[1]:5d} launches {len(dur[k]):3d} avg {us:8.1f} us | GRBM_GUI_ACTIVE {g:12.0f} -> /8 XCD /us = {g / 8 / us / 1e3:5.3f} GHz (raw/us {g / us / 1e3:6.3f}) | GRBM_COUNT {n:12.0f} -> /8/us {n / 8 / us / 1e3:5.3f} | SQ_BUSY_CYCLES {s:12.0f} -> /32 SE /us = {s / 32 / us / 1e3:5.3f} GHz")
PY
  rm -rf gpurun_out/pmcc
}
echo "--- 2/4. pmc passes"
CLOCK_RECON_NO_SMI=1 pmc_pass probe tools/probes/clock_recon.bin 0.15
pmc_pass gate_up python3 tools/gemm_one.py 768 37888 3584
pmc_pass down_proj python3 tools/gemm_one.py 768 3584 19968
pmc_pass hadamard python3 tools/had_one.py
cat $O/pmc.txt
echo "--- 6. ws timeline"; MQUANT_HIP_LIB=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_wstl.so timeout 600 python3 tools/gemm_timeline.py > $O/ws_timeline.txt 2>&1; echo "rc=$?"; tail -20 $O/ws_timeline.txt
echo "--- 7. kslope with stamps"; MQ_STAMPS=1 timeout 300 python3 tools/gemm_kslope.py --tile 14 --libs ppst > $O/kslope_pp.txt 2>&1; cat $O/kslope_pp.txt
