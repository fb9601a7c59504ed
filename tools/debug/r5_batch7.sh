cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rope_gemm.py tests/test_gpu_checkpoint.py tests/test_gpu_prefill_objects.py tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/r5_tests5.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r5_tests5.log
MQUANT_HIP_LIB=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_wstl.so timeout 600 python3 tools/gemm_timeline.py > gpurun_out/r5_ws_timeline3.txt 2>&1; echo "tl rc=$?"; grep -E "^==|entry|first instruction" gpurun_out/r5_ws_timeline3.txt | cut -c1-200
timeout 600 bash tools/full_prefill_prof.sh > gpurun_out/r5_full_prefill_kernel_split.txt 2>&1; echo "split rc=$?"; head -40 gpurun_out/r5_full_prefill_kernel_split.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5_bench_rope.json 2> gpurun_out/r5_bench_rope.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r5_bench_rope.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['step_frac'], d['full_prefill'])"
