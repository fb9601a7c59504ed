cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" _sku4; do echo "== lib$v round $r (main: 8 k-tiles in flight per wave at M <= 16; sku4: 4)"; MQUANT_HIP_LIB=mquant_amd/libmquant_hip$v.so timeout 600 python3 tools/decode_gemm_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done; done | tee gpurun_out/r5_decode_skinny_inflight_ab.txt
