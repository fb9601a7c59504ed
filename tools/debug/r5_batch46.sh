cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py 2> gpurun_out/r5_final_bench.err | tail -1 > gpurun_out/r5_final_bench.json
python3 -c "
import json;d=json.load(open('gpurun_out/r5_final_bench.json'));r=d['roofline'];print(d['value'],d['ms_per_step'],r['frac'],r['step_frac'],r.get('traffic_stale'),r.get('traffic'),r['peak_sustained_measured']['value'],r['peak_sustained_measured']['frac_of_sustained'],d['full_prefill']['ttft_ms_median'],d['cpu_baseline']['value'])"
for w in qwenvl_7b internvl2_8b qwen2vl_72b; do
  B=1; [ $w = internvl2_8b ] && B=4
  timeout 1500 python3 bench.py --workload $w --batch $B --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_bench_$w.json
  python3 -c "
import json;d=json.load(open('gpurun_out/r5_bench_$w.json'));r=d['roofline'];print('$w',d['value'],d['ms_per_step'],r['frac'],r['step_frac'],r.get('gemm_ms_per_step'))"
done
