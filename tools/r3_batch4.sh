#!/bin/bash
# round-3: full GPU test suite + the secondary workloads (BASELINE configs 2, 4, 5)
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3d
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r3d/tests_all.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3d/tests_all.log
for w in qwenvl_7b "internvl2_8b --batch 4" qwen2vl_72b; do
  name=$(echo $w | cut -d' ' -f1)
  timeout 1200 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r3d/bench_$name.json 2> gpurun_out/r3d/bench_$name.err; echo "$name rc=$?"; tail -2 gpurun_out/r3d/bench_$name.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3d/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, j["value"], j["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "gemm", r["gemm_ms_per_step"], "quant", r["quant_hadamard_ms_per_step"], "launches", r["launches_per_step"], "TOP", j["config"]["gemm_TOP_per_step"], "GB", j["config"]["weights_GB"])
    except Exception as e:
        print(f, "ERR", e)
PY
