#!/bin/bash
# usage: tools/traffic_prof.sh <out.json> <commit>  -> HBM bytes per launch of our kernels from two
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, kernels serialised) over bench.py
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=$1; COMMIT=$2
CMD="python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-full-prefill --no-secondary --no-floor-model"
rm -rf gpurun_out/tr; mkdir -p gpurun_out/tr $(dirname $OUT)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/tr -o f -- $CMD > gpurun_out/tr/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/tr -o w -- $CMD > gpurun_out/tr/w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/tr/f_counter_collection.csv gpurun_out/tr/w_counter_collection.csv $OUT "$COMMIT" \
    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $CMD"
rm -f gpurun_out/tr/*_counter_collection.csv gpurun_out/tr/*_kernel_trace.csv
