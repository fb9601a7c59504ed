#!/bin/bash
# usage: tools/gemm_pmc.sh <shape> <configs> -> SQ counter summary (rocprofv3 --pmc, kernels serialised) for one shape
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc -o t -- python3 tools/gemm_bench.py --only "$1" --configs="$2" > gpurun_out/pmc/log 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/pmc/t_counter_collection.csv | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    us=float(r['avg_us']); clk=float(r['SQ_BUSY_CYCLES'])/32/us/1e3
    print(f\"{r['kernel'][:52]:52s} wg {r['workgroups']:>4s} {us:7.1f} us clk {clk:4.2f} GHz mfma_busy {float(r['SQ_VALU_MFMA_BUSY_CYCLES'])/(us*clk*1e3*1024):.3f} valu/mfma {r['valu_per_mfma']} wait {r['wait_any']} issue_stall {r['wait_inst_any']} active {r['active_inst_any']}\")
"
rm -f gpurun_out/pmc/t_counter_collection.csv gpurun_out/pmc/t_kernel_trace.csv
