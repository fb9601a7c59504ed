cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5clock
timeout 600 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_rank2.py -x -q -m gpu > gpurun_out/r5_tests2.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_tests2.log
timeout 900 python3 tools/r5_ws_mfma_ab.py > gpurun_out/r5_ws_mfma_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r5_ws_mfma_ab.txt
bash tools/r5_clock2.sh > gpurun_out/r5clock/clock2.log 2>&1; tail -30 gpurun_out/r5clock/pmc2.txt
for v in "" "--direct-engines"; do timeout 600 python bench.py --workload qwenvl_7b --no-cpu-baseline --no-full-prefill $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('qwenvl_7b', '$v', 'tok/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'gemm_ms', round(r['gemm_ms_per_step'],3), 'frac', round(r['frac'],4), 'launches', r.get('launches_per_step'), 'quant_had_ms', round(r['quant_hadamard_ms_per_step'],3))"; done > gpurun_out/r5_qwenvl_paths.txt 2>&1; cat gpurun_out/r5_qwenvl_paths.txt
