cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgroup.py tests/test_gpu_groupwise.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/wgroup_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_group_gemm_ws_fold_final.txt
timeout 600 python3 bench.py 2>&1 | tail -1 > gpurun_out/r5_bench_after_group_fold.json
