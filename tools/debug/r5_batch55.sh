cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgroup.py tests/test_gpu_checkpoint.py tests/test_gpu_gptq.py -x -q -m gpu 2>&1 | tail -15
