cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_tests6.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_tests6.log
bash tools/bench_ab.sh "main lazy mathfirst" 3 > gpurun_out/r5_bench_ab_args_and_wave_order.txt 2>&1; cat gpurun_out/r5_bench_ab_args_and_wave_order.txt
MQUANT_HIP_LIB=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_wstl.so timeout 600 python3 tools/gemm_timeline.py --shapes vit_proj,llm_qkv,llm_down > gpurun_out/r5_ws_timeline4.txt 2>&1; grep -E "^==|entry|first instruction|launch:" gpurun_out/r5_ws_timeline4.txt | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5_bench_b8.json 2> gpurun_out/r5_bench_b8.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r5_bench_b8.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['step_frac'], r['gemm_ms_per_step'], r['quant_hadamard_ms_per_step'], d['full_prefill']['ttft_ms_median'])"
