#!/bin/bash
# usage: tools/gemm_prof.sh <shape-filter> <configs>   -> per-kernel rocprof durations for tile configs
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gp; mkdir -p gpurun_out/gp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -o t -- python3 tools/gemm_bench.py --only "$1" --configs="$2" --bits ${3:-4} > gpurun_out/gp/log 2>&1
python3 tools/trace_summary.py gpurun_out/gp/t_kernel_trace.csv | { grep "gemm\|splitk" || true; }
rm -f gpurun_out/gp/t_kernel_trace.csv
