#!/bin/bash
# round 4: ping-pong 256x256 kernel (tile 14) against the pipelined one (13) on the wide shapes, cold weights
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_pp_tests.txt
for s in "llm.gate_up" "72b.gate_up" "ivl.w1w3" "qvl.w1w2" "ivl4.w1w3" "llm.down"; do
  python tools/gemm_bench.py --only "$s" --configs=-1:0,13:1,14:1,13:1,14:1 --tiled --cold 2>&1 | tail -2
done > gpurun_out/r4_pp_ab.txt 2>&1
cat gpurun_out/r4_pp_tests.txt gpurun_out/r4_pp_ab.txt
