cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in -1 45 47 48; do echo "== forced tile $t (-1: the plan)"; timeout 600 python3 tools/wgroup_bench.py --tile $t --modes w,wx 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5_group_gemm_ws_fold_tiles2.txt
