#!/bin/bash
# round-5 profile artefacts of the default bench command: kernel stats (rocprofv3 --kernel-trace --stats), SQ counters,
# HBM traffic (PMC passes), and the plain bench lines.  Everything lands in gpurun_out/r5p/ (copied to profiles/ by hand).
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
COMMIT=${1:-unknown}
mkdir -p gpurun_out/r5p
timeout 900 python3 bench.py > gpurun_out/r5p/r5_final_bench.json 2> gpurun_out/r5p/bench.err; echo "bench rc=$?"
timeout 600 python3 bench.py --had-fast --no-cpu-baseline --no-full-prefill > gpurun_out/r5p/r5_bench_had_fast.json 2>> gpurun_out/r5p/bench.err; echo "fast rc=$?"
timeout 600 python3 bench.py --no-fuse --no-cpu-baseline --no-full-prefill > gpurun_out/r5p/r5_bench_no_fuse.json 2>> gpurun_out/r5p/bench.err; echo "nofuse rc=$?"
timeout 600 python3 bench.py --batch 8 --no-cpu-baseline --no-full-prefill > gpurun_out/r5p/r5_bench_batch8.json 2>> gpurun_out/r5p/bench.err; echo "b8 rc=$?"
timeout 900 tools/bench_prof.sh gpurun_out/r5p/r5_final_prof > gpurun_out/r5p/bench_prof.log 2>&1; mv gpurun_out/r5p/r5_final_prof_kernel_stats.csv gpurun_out/r5p/r5_final_kernel_stats.csv; mv gpurun_out/r5p/r5_final_prof_bench.json gpurun_out/r5p/r5_final_bench_under_rocprof.json; echo "prof rc=$?"
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/sq -o s -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-full-prefill > gpurun_out/sq/log 2>&1; echo "sq rc=$?"
python3 tools/pmc_sq_summary.py gpurun_out/sq/s_counter_collection.csv > gpurun_out/r5p/r5_final_sq_counters.csv; rm -rf gpurun_out/sq
timeout 1200 tools/traffic_prof.sh gpurun_out/r5p/r5_traffic.json "$COMMIT" > gpurun_out/r5p/traffic.log 2>&1; echo "traffic rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5p/r5_*bench*.json")):
    if "rocprof" in f: continue
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, j["value"], j["ms_per_step"], "frac", r["frac"], "step_frac", r["step_frac"], "gemm", r["gemm_ms_per_step"], "quant", r["quant_hadamard_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
head -30 gpurun_out/r5p/r5_final_kernel_stats.csv
