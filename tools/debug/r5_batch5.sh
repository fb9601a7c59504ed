cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_tests_full.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_tests_full.log
bash tools/bench_ab.sh "main ws32" 3 > gpurun_out/r5_bench_ab_ws16_vs_ws32.txt 2>&1; cat gpurun_out/r5_bench_ab_ws16_vs_ws32.txt
timeout 600 python3 tools/r5_ws_mfma_ab.py --tiles 14,46,42 --kinds bench-like --shapes ivl4_w2,gate_up,qvl_c_attn,72b_qkv > gpurun_out/r5_pp_vs_ws192.txt 2>&1; cat gpurun_out/r5_pp_vs_ws192.txt
timeout 600 python bench.py > gpurun_out/r5_bench_ws16.json 2> gpurun_out/r5_bench_ws16.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r5_bench_ws16.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['step_frac'], r.get('peak_sustained_measured'), d['full_prefill']['ttft_ms_median'])"
