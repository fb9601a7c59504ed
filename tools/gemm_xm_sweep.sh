#!/bin/bash
# usage: tools/gemm_xm_sweep.sh -> per model shape, heuristic tile with the m-grouping of the XCD map forced to 1, 2, 4 (and auto)
set -euo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for xm in 0 256 512 1024; do
  echo "#### splits word $xm (xm = $((xm >> 8)), 0 = automatic)"
  tools/gemm_prof_shapes.sh "-1:$xm" 4 --tiled
done
