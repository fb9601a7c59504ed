cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_dynamic.py -x -q -m gpu > gpurun_out/r5_tests10.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_tests10.log
timeout 900 python3 tools/r5_ws_mfma_ab.py --tiles 47,49,48,45 --kinds bench-like --shapes vit_proj,vit_fc2,vit_qkv,llm_o > gpurun_out/r5_ws_tile49_ab.txt 2>&1; cat gpurun_out/r5_ws_tile49_ab.txt
