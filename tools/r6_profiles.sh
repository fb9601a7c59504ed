#!/bin/bash
# round-6 profile artefacts of the default bench command: the plain bench line, kernel stats (rocprofv3 --kernel-trace --stats), SQ
# counters, HBM traffic (PMC passes), the per-kernel split of one whole prefill, host overhead of the eager wrapper.  Everything lands
# in gpurun_out/r6p/ (copied to profiles/ by hand).  usage: tools/r6_profiles.sh <commit>
set -uo pipefail
GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
COMMIT=${1:-unknown}
O=gpurun_out/r6p
mkdir -p $O
timeout 900 python3 bench.py > $O/r6_final_bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 tools/bench_prof.sh $O/r6_final_prof > $O/bench_prof.log 2>&1; echo "prof rc=$?"
mv $O/r6_final_prof_kernel_stats.csv $O/r6_final_kernel_stats.csv; mv $O/r6_final_prof_bench.json $O/r6_final_bench_under_rocprof.json
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/sq -o s -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-full-prefill --no-secondary --no-floor-model > gpurun_out/sq/log 2>&1; echo "sq rc=$?"
python3 tools/pmc_sq_summary.py gpurun_out/sq/s_counter_collection.csv > $O/r6_final_sq_counters.csv; rm -rf gpurun_out/sq
[ -n "${SKIP_TRAFFIC:-}" ] || { timeout 1200 tools/traffic_prof.sh $O/r6_traffic.json "$COMMIT" > $O/traffic.log 2>&1; echo "traffic rc=$?"; }
timeout 600 tools/full_prefill_prof.sh > $O/r6_full_prefill_kernel_split.txt 2>&1; echo "fp rc=$?"
timeout 300 python3 tools/decode_host_overhead.py > $O/r6_eager_host_overhead.txt 2>&1; echo "host rc=$?"
timeout 600 python3 bench.py --had-fast --no-cpu-baseline --no-full-prefill --no-secondary > $O/r6_bench_had_fast.json 2>> $O/bench.err; echo "fast rc=$?"
head -40 $O/r6_final_kernel_stats.csv
tail -30 $O/r6_full_prefill_kernel_split.txt
cat $O/r6_eager_host_overhead.txt
