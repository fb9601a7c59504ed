cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python3 tools/decode_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_decode_gemm_bench_final.txt
timeout 900 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_bench_after_skinny.json
python3 -c "
import json;d=json.load(open('gpurun_out/r5_bench_after_skinny.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('step_frac'),d.get('full_prefill',{}).get('ttft_ms_median'))"
