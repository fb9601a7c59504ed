cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "" _wgabl1 _wgabl2 _wgabl3; do
  echo "== lib libmquant_hip$v.so (ablation: ${v:-none}; 1 = no scale reloads, 2 = no fold arithmetic, 3 = both)"
  MQUANT_HIP_LIB=mquant_amd/libmquant_hip$v.so timeout 600 python3 tools/wgroup_bench.py --shapes llm.qkv,llm.o,llm.down 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5_group_fold_ablations.txt
