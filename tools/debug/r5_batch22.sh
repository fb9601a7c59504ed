cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "" _wgabl4 _wgabl8 _wgabl12; do
  echo "== lib libmquant_hip$v.so (timing-only ablations: 4 = no fold intervals, 8 = no scale DMAs, 12 = both)"
  MQUANT_HIP_LIB=mquant_amd/libmquant_hip$v.so timeout 600 python3 tools/wgroup_bench.py --shapes llm.qkv,llm.o,llm.down --modes w 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5_group_fold_ablations2.txt
