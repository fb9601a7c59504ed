cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hadamard_fast.py tests/test_gpu_full_size.py -x -q -m gpu -k "had or Had" 2>&1 | tail -3
for r in 1 2 3; do for v in main hadsep; do
  lib=mquant_amd/libmquant_hip_$v.so; [ "$v" = main ] && lib=mquant_amd/libmquant_hip.so
  echo "== $v round $r"; MQUANT_HIP_LIB=$lib timeout 300 python3 tools/had_bench.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r5_hadamard_mask_loads_ab.txt
bash tools/bench_ab.sh "main hadsep" 3 | tee gpurun_out/r5_bench_ab_hadamard_mask_loads.txt
