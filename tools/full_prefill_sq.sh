#!/bin/bash
# usage: tools/full_prefill_sq.sh <out.csv> -> SQ counters per kernel of the whole synthetic prefill (FullPrefill.step, fused glue): MFMA utilisation, vector
# instructions per MFMA, wait fractions -- one rocprofv3 --pmc pass (counters only, kernels serialised), summarised by tools/pmc_sq_summary.py
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r6_full_prefill_sq_counters.csv}
rm -rf gpurun_out/fpsq; mkdir -p gpurun_out/fpsq $(dirname $OUT)
cat > gpurun_out/fpsq/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
torch.set_grad_enabled(False)
from mquant_amd import workload
from mquant_amd.full_prefill import FullPrefill
dev = torch.device("cuda:0")
pf = workload.Prefill(workload.qwen2vl_7b_specs(msq=True), device=dev, share_groups=True)
fp = FullPrefill(pf, fused_glue=True)
fp.calibrate()
for _ in range(6):
    fp.step()
torch.cuda.synchronize()
PY
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/fpsq -o s -- python3 gpurun_out/fpsq/run.py > gpurun_out/fpsq/log 2>&1; echo "sq rc=$?"
python3 tools/pmc_sq_summary.py gpurun_out/fpsq/s_counter_collection.csv > $OUT; rm -rf gpurun_out/fpsq
grep -E "attn_prefill|rmsn|rope|gemv|kernel,work" $OUT | cut -c1-230
