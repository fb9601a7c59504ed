cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/r5_profiles.sh 652c025 > gpurun_out/r5_profiles.log 2>&1; tail -42 gpurun_out/r5_profiles.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/r5p/r5_final_bench_after_traffic.json 2> gpurun_out/r5p/err_after; python3 -c "
import json; d=json.loads(open('gpurun_out/r5p/r5_final_bench_after_traffic.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['step_frac'], r['traffic_stale'], r['traffic'], r['peak_sustained_measured']['value'], d['full_prefill']['ttft_ms_median'], d['full_prefill']['ttft_ms_median_rope_as_its_own_launch'])"
