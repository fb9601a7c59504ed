cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/r5_tests3.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_tests3.log
timeout 1200 python3 tools/r5_ws_mfma_ab.py --tiles 40,44,48,41,45,42,46,43,47 --kinds bench-like --shapes llm_down,llm_o,llm_qkv,qvl_down,qvl_c_attn,72b_o,72b_qkv,72b_down,ivl_w2,ivl_wqkv,ivl4_w2,vit_proj,vit_fc2,vit_qkv,vit_fc1 > gpurun_out/r5_ws_tiles_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r5_ws_tiles_ab.txt
