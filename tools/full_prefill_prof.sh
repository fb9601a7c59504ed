#!/bin/bash
# usage: tools/full_prefill_prof.sh -> per-kernel time of ONE whole synthetic prefill (FullPrefill.step, fused glue), rocprofv3 --kernel-trace
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fp; mkdir -p gpurun_out/fp
cat > gpurun_out/fp/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
torch.set_grad_enabled(False)
from mquant_amd import workload
from mquant_amd.full_prefill import FullPrefill
dev = torch.device("cuda:0")
pf = workload.Prefill(workload.qwen2vl_7b_specs(msq=True), device=dev, share_groups=True)
fp = FullPrefill(pf, fused_glue=True)
fp.calibrate()
for _ in range(3):
    fp.step()
torch.cuda.synchronize()
print("MARK_BEGIN", flush=True)
for _ in range(5):
    fp.step()
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fp -o t -- python3 gpurun_out/fp/run.py > gpurun_out/fp/log 2>&1
python3 - <<'PY'
import collections, csv
rows = list(csv.DictReader(open("gpurun_out/fp/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 5 steps: find the repeating period by counting launches of the lm_head-sized tail; simpler: take the last 5/8 of steady-state launches
n = len(rows)
# calibration + 3 warm-up + 5 timed steps; a step has a fixed number of launches L: steady part = last 8 L launches (roughly); find L from the
# positions of the rms_norm of the last position (unique per step): use the Hadamard down_proj kernel count instead (28 per step)
idx = [i for i, r in enumerate(rows) if "hadamard_kernel" in r["Kernel_Name"] and ", 5>" in r["Kernel_Name"]]   # the decoder's down_proj rotation (5 x 2 units)
per_step = 28
last = idx[-5 * per_step]            # first down_proj Hadamard of the 5 timed steps
# back up to the start of that step: the first kernel after the previous step's last launch -> approximate by the launch count of a step
L = (idx[-1] - idx[-per_step - 1]) * 1  # launches between the last Hadamard of step k-1 and of step k = one step
start = len(rows) - 5 * L
agg = collections.OrderedDict()
for r in rows[start:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    name = name[:90]
    agg.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"launches per step {L}, kernel time per step {tot / 5 / 1e6:.3f} ms")
print("kernel,calls_per_step,avg_us,ms_per_step,share")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{k}",{len(v) / 5:.1f},{sum(v) / len(v) / 1e3:.2f},{sum(v) / 5 / 1e6:.3f},{sum(v) / tot:.4f}')
PY
rm -f gpurun_out/fp/t_kernel_trace.csv
