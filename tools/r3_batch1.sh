#!/bin/bash
# round-3 first GPU pass: new parity tests, then the bench lines (wrapper-built default, direct engines, unfused)
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_prefill_objects.py tests/test_gpu_toy_pipeline.py tests/test_gpu_dynamic.py -q -m gpu > gpurun_out/r3a/tests1.log 2>&1; echo "tests1 rc=$?" 
tail -15 gpurun_out/r3a/tests1.log
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r3a/tests_multi.log 2>&1; echo "multi rc=$?"
tail -8 gpurun_out/r3a/tests_multi.log
timeout 900 python bench.py > gpurun_out/r3a/bench_wrappers.json 2> gpurun_out/r3a/bench_wrappers.err; echo "bench rc=$?"; tail -3 gpurun_out/r3a/bench_wrappers.err
timeout 600 python bench.py --direct-engines --no-cpu-baseline --no-full-prefill > gpurun_out/r3a/bench_direct.json 2> gpurun_out/r3a/bench_direct.err; echo "direct rc=$?"
timeout 600 python bench.py --no-fuse --no-cpu-baseline --no-full-prefill > gpurun_out/r3a/bench_nofuse.json 2> gpurun_out/r3a/bench_nofuse.err; echo "nofuse rc=$?"
timeout 600 python bench.py --no-logits --no-cpu-baseline --no-full-prefill > gpurun_out/r3a/bench_nologits.json 2> gpurun_out/r3a/bench_nologits.err; echo "nologits rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j["roofline"]
        print(f, j["value"], j["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "gemm", r["gemm_ms_per_step"], "quant", r["quant_hadamard_ms_per_step"], "hot", r["hot_path_ms_per_step_stream_timed"], "launches", r["launches_per_step"], j["config"].get("lm_head_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
