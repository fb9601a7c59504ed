#!/bin/bash
# round 4: ping-pong tiles (14..19) against the shipped plan on every model shape, cold weights
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4_pp_tests.txt
{
python tools/gemm_bench.py --only "llm.down" --configs=-1:0,16:1,15:1,17:1,16:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "llm.qkv" --configs=-1:0,15:1,16:1,17:1,15:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "llm.q/o" --configs=-1:0,16:1,15:1,18:1,16:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "vis.qkv" --configs=-1:0,15:1,16:1,17:1,18:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "vis.proj" --configs=-1:0,18:1,16:1,15:1,18:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "vis.fc1" --configs=-1:0,17:1,15:1,16:1,19:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "vis.fc2" --configs=-1:0,18:1,16:1,15:1,18:1 --tiled --cold 2>&1 | tail -2
python tools/gemm_bench.py --only "llm.gate_up" --configs=13:1,14:1,19:1,17:1,14:1 --tiled --cold 2>&1 | tail -2
} > gpurun_out/r4_pp_shapes.txt 2>&1
cat gpurun_out/r4_pp_tests.txt gpurun_out/r4_pp_shapes.txt
