cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/r5_tests7.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r5_tests7.log
bash tools/r5_profiles.sh 652c025 > gpurun_out/r5_profiles.log 2>&1; tail -45 gpurun_out/r5_profiles.log
