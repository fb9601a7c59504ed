cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
timeout 900 python bench.py > gpurun_out/r5s/r5_final_bench_b.json 2> gpurun_out/r5s/err0; echo "bench rc=$?"
timeout 900 python bench.py --workload qwenvl_7b --no-cpu-baseline > gpurun_out/r5s/r5_bench_qwenvl_7b.json 2> gpurun_out/r5s/err1; echo "qwenvl rc=$?"
timeout 900 python bench.py --workload internvl2_8b --batch 4 --no-cpu-baseline > gpurun_out/r5s/r5_bench_internvl2_8b.json 2> gpurun_out/r5s/err2; echo "ivl rc=$?"
timeout 1200 python bench.py --workload qwen2vl_72b --no-cpu-baseline > gpurun_out/r5s/r5_bench_qwen2vl_72b.json 2> gpurun_out/r5s/err3; echo "72b rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5s/r5_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "gemm", r["gemm_ms_per_step"], "quant", r["quant_hadamard_ms_per_step"], "stale", r.get("traffic_stale"), "sust", (r.get("peak_sustained_measured") or {}).get("value"), (d.get("full_prefill") or {}).get("ttft_ms_median"), (d.get("full_prefill") or {}).get("ttft_ms_median_rope_as_its_own_launch"))
    except Exception as e:
        print(f, "ERR", e)
PY
