#!/bin/bash
# round 4: vector-ALU exact Hadamard kernel -- parity tests, then time against the matrix-core exact kernel
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_hadamard_valu.py tests/test_gpu_kernels.py tests/test_gpu_act_hadamard.py tests/test_gpu_hadamard_fast.py -x -q -m gpu 2>&1 | tail -8
for i in 0 1 0 1; do HAD_IMPL=$i HAD_SHAPES=vis.fc2,llm.down python tools/had_bench.py 2>&1 | tail -3; done
