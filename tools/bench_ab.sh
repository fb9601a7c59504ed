#!/bin/bash
# usage: tools/bench_ab.sh "<variants>" [rounds] [extra bench flags] -> same-box A/B of whole-bench lines for library variants
# (mquant_amd/libmquant_hip_<v>.so; "main" = the regular build), alternating, hot path only
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_$v.so; [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip.so
    MQUANT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-full-prefill --no-secondary ${3:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'round $r', 'tok/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'gemm_ms', round(r['gemm_ms_per_step'],3), 'frac', round(r['frac'],4), 'step_frac', round(r['step_frac'],4), 'quant_had_ms', round(r['quant_hadamard_ms_per_step'],3))"
  done
done
