cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" _foldfma; do
  echo "== lib libmquant_hip$v.so round $r"
  MQUANT_HIP_LIB=mquant_amd/libmquant_hip$v.so timeout 600 python3 tools/wgroup_bench.py --shapes llm.qkv,llm.o,llm.gate_up,llm.down 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r5_group_fold_fma_timing.txt
