cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgroup.py tests/test_gpu_checkpoint.py tests/test_gpu_tiled.py tests/test_gpu_groupwise.py -x -q -m gpu 2>&1 | tail -12
timeout 600 python3 tools/wgroup_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_group_gemm_ws_fold2.txt
for t in 45 47 48; do echo "== forced tile $t"; timeout 600 python3 tools/wgroup_bench.py --tile $t --shapes llm.qkv,llm.o,llm.down,vit.fc1 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5_group_gemm_ws_fold_tiles.txt
timeout 600 python3 tools/w8_image_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_w8_image_ab.txt
