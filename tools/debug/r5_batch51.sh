cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_kernels.py tests/test_gpu_skinny.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 tools/midm_gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep "llm.down" | tee gpurun_out/r5_midm_gemm_bench_after.txt
