cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --w-groupsize 128 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_bench_wgroup128.json
timeout 900 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_bench_same_box_default.json
timeout 1500 python3 bench.py --workload qwen2vl_72b --w-groupsize 128 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_bench_qwen2vl_72b_wgroup128.json
timeout 1500 python3 bench.py --workload qwen2vl_72b --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_bench_qwen2vl_72b_same_box.json
for f in r5_bench_wgroup128 r5_bench_same_box_default r5_bench_qwen2vl_72b_wgroup128 r5_bench_qwen2vl_72b_same_box; do python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'), d['config']['path'][:60])
except Exception as e: print('$f', 'ERR', e, open('gpurun_out/$f.json').read()[-600:])
"; done
