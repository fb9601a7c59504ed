cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hadamard_fast.py tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | tail -3
HAD_ROWS=1,4,16,64,130,256 HAD_SHAPES=vis.fc2,llm.down timeout 300 python3 tools/had_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_hadamard_small_m_after.txt
timeout 600 python3 tools/decode_step_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_decode_step_bench.txt
