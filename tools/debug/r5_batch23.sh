cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "" _foldscalar; do
  echo "== lib libmquant_hip$v.so (foldscalar: the fold with scalar fp32 instructions, -fno-slp-vectorize)"
  MQUANT_HIP_LIB=mquant_amd/libmquant_hip$v.so timeout 600 python3 tools/wgroup_bench.py --shapes llm.qkv,llm.o,llm.down,vit.fc1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5_group_fold_packed_vs_scalar.txt
MQUANT_HIP_LIB=mquant_amd/libmquant_hip_foldscalar.so timeout 900 python -m pytest tests/test_gpu_wgroup.py tests/test_gpu_groupwise.py -x -q -m gpu 2>&1 | tail -3
