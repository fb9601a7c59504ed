cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_tests9.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r5_tests9.log
