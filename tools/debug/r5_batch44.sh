cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_skinny.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/decode_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_decode_gemm_bench_wg.txt
timeout 600 python3 tools/decode_step_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_decode_step_bench_wg.txt
