cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MQUANT_HIP_LIB=mquant_amd/libmquant_hip_sknofence.so timeout 600 python3 tools/decode_gemm_bench.py 2>&1 | grep -v amdgpu.ids
