#!/bin/bash
# round-3: fast Hadamard mode -- exact kernel still bit-exact after the refactor, fast-mode tests, flip rates, timings
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests/test_gpu_hadamard_fast.py tests/test_gpu_kernels.py tests/test_gpu_act_hadamard.py tests/test_gpu_wrapper_golden.py tests/test_gpu_prefill_objects.py tests/test_gpu_toy_pipeline.py tests/test_gpu_full_size.py -q -m gpu > gpurun_out/r3b/tests.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r3b/tests.log
timeout 600 python tools/had_fast_flips.py > gpurun_out/r3b/had_fast_flips.txt 2>&1; echo "flips rc=$?"; cat gpurun_out/r3b/had_fast_flips.txt | grep -v amdgpu.ids
HAD_FAST=0 timeout 300 python tools/had_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3b/had_bench_exact.txt; HAD_FAST=1 timeout 300 python tools/had_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3b/had_bench_fast.txt
paste -d'\n' gpurun_out/r3b/had_bench_exact.txt gpurun_out/r3b/had_bench_fast.txt | grep float16
timeout 600 python bench.py --had-fast --no-cpu-baseline --no-full-prefill > gpurun_out/r3b/bench_had_fast.json 2> gpurun_out/r3b/bench_had_fast.err; echo "bench fast rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-full-prefill > gpurun_out/r3b/bench_exact.json 2> gpurun_out/r3b/bench_exact.err; echo "bench exact rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, j["value"], j["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "gemm", r["gemm_ms_per_step"], "quant", r["quant_hadamard_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
