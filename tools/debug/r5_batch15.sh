cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wgroup.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/wgroup_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_wgroup_bench.txt
