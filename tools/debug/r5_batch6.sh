cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgroup.py tests/test_gpu_checkpoint.py tests/test_gpu_groupwise.py -x -q -m gpu > gpurun_out/r5_tests4.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r5_tests4.log
MQUANT_HIP_LIB=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_wstl.so timeout 600 python3 tools/gemm_timeline.py > gpurun_out/r5_ws_timeline2.txt 2>&1; echo "tl rc=$?"; grep -E "^==|entry ->|   ->|requested ->" gpurun_out/r5_ws_timeline2.txt | cut -c1-200
