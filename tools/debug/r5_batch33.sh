cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python3 tools/decode_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_decode_gemm_bench_split.txt
