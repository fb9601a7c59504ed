#!/bin/bash
# usage: tools/ttft_ab.sh "<variants>" [rounds] -> same-box A/B of the whole-prefill TTFT (bench.py's full_prefill block) for library
# variants (mquant_amd/libmquant_hip_<v>.so; "main" = the regular build), alternating
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip_$v.so; [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/mquant_amd/libmquant_hip.so
    MQUANT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --no-floor-model 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['full_prefill']
print('$v', 'round $r', 'tok/s', round(d['value']), 'frac', d['roofline']['frac'], 'ttft', f['ttft_ms_median'], 'p90', f['ttft_ms_p90'], 'prologue form', f.get('ttft_ms_median_act_in_hadamard_prologue'), 'unfused', f.get('ttft_ms_median_unfused_glue'))"
  done
done
