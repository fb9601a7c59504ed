#!/bin/bash
# usage: tools/gemm_pmc_any.sh <shape> <configs> "<counter list>" [extra gemm_bench flags] -> per-kernel averages of the named counters
# (rocprofv3 --pmc with --kernel-trace only; one pass per call)
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
timeout 240 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d gpurun_out/pmc -o t -- python3 tools/gemm_bench.py --only "$1" --configs="$2" --tiled ${4:-} > gpurun_out/pmc/log 2>&1 || { tail -5 gpurun_out/pmc/log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc/t_counter_collection.csv gemm
rm -f gpurun_out/pmc/t_counter_collection.csv gpurun_out/pmc/t_kernel_trace.csv
