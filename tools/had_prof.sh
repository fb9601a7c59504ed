#!/bin/bash
# per-kernel rocprof durations of a Hadamard script (default tools/had_bench.py)
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hp; mkdir -p gpurun_out/hp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hp -o t -- python3 ${1:-tools/had_bench.py} > gpurun_out/hp/log 2>&1
{ grep -v "^W2026\|^E2026" gpurun_out/hp/log || true; } | tail -20
python3 tools/trace_summary.py gpurun_out/hp/t_kernel_trace.csv | { grep "hadamard" || true; }
rm -f gpurun_out/hp/t_kernel_trace.csv
