#!/bin/bash
# Round 5, second clock pass: the counter readings (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES over the kernel's duration) CALIBRATED on launches
# whose clock the in-kernel stamps pin (tools/probes/clock_recon.bin: 2.39 GHz on zeros, 1.63 GHz on random bytes), then taken on the
# product's gate|up (ping-pong) and down_proj (ws) launches.  Output: gpurun_out/r5clock/pmc2.txt
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5clock; mkdir -p $O; rm -f $O/pmc2.txt
export CLOCK_RECON_NO_SMI=1
pmc_pass() {   # name, command...
  local name=$1; shift
  rm -rf gpurun_out/pmcc; mkdir -p gpurun_out/pmcc
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcc -o t -- "$@" > gpurun_out/pmcc/log 2>&1 || { echo "pmc pass $name failed"; tail -5 gpurun_out/pmcc/log; }
  grep -E "^== burn|meter" gpurun_out/pmcc/log | cut -c1-260 >> $O/pmc2.txt
  python3 - "$name" <<'PY' >> gpurun_out/r5clock/pmc2.txt
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
    if us < 30 or "at::" in k[0] or "prepack" in k[0]: continue
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    g, s = avg.get("GRBM_GUI_ACTIVE", 0), avg.get("SQ_BUSY_CYCLES", 0)
    print(f"  {k[0]:60s} wg {k[1]:5d} launches {len(dur[k]):3d} avg {us:8.1f} us | GRBM_GUI_ACTIVE {g:12.0f} /8 XCD /us = {g / 8 / us / 1e3:5.3f} GHz | SQ_BUSY_CYCLES {s:12.0f} /32 SE /us = {s / 32 / us / 1e3:5.3f} GHz")
PY
  rm -rf gpurun_out/pmcc
}
pmc_pass "probe, 32x32x32 zeros (in-kernel clock in the line above)" tools/probes/clock_recon.bin 0.3 "32x32x32 zeros"
pmc_pass "probe, 32x32x32 random" tools/probes/clock_recon.bin 0.3 "32x32x32 random"
pmc_pass "probe, 16x16x64 random" tools/probes/clock_recon.bin 0.3 "16x16x64 random"
pmc_pass "gate|up 768 x 37888 x 3584 (ping-pong, tiled activations, random bytes)" python3 tools/gemm_one.py 768 37888 3584
pmc_pass "down_proj 768 x 3584 x 19968 (ws 96 x 128)" python3 tools/gemm_one.py 768 3584 19968
cat $O/pmc2.txt
