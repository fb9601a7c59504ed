#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_hadamard_valu.py -x -q -m gpu 2>&1 | tail -3
for i in 0 1 0 1; do HAD_IMPL=$i HAD_SHAPES=vis.fc2,llm.down python tools/had_bench.py 2>&1 | tail -2; done
export HAD_IMPL=0
bash tools/had_pmc.sh 0 llm.down,vis.fc2 2>&1 | tail -60
